#!/bin/bash
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
PASSES=(
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_BUSY_CU_CYCLES"
 "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_LDS_UNALIGNED_STALL SQC_TC_STALL SQC_DCACHE_MISSES SQC_DCACHE_REQ"
)
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --streams 1 > "$OUT/pass$i.log" 2>&1
  i=$((i+1))
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
grep -A30 "k_screen_encode" "$OUT/summary.txt" | head -34
