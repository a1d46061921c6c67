#!/bin/bash
# GPU box helper (round 3): PMC passes (counters only) over tools/pipeline_diag.py for one pipeline.
# usage: tools/r3_pmc.sh <outdir> <tile|launches> [frames]
set -e
OUT=$1; export MI355_JPEG_PIPELINE=$2; N=${3:-64}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
 "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_SMEM"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum"
)
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 tools/pipeline_diag.py $N > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed: $P"
  i=$((i+1))
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
find "$OUT" -name "*.csv" -size +1M -delete
cat "$OUT/summary.txt"
