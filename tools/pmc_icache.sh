#!/bin/bash
# GPU box helper: instruction-cache counters of the bench's kernels (one rocprofv3 process per pass, counters only).
# usage: tools/pmc_icache.sh <outdir>
set -e
OUT=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
rocprofv3 -L > "$OUT/counters.txt" 2>&1 || true
grep -i -o "SQC_[A-Z0-9_]*\|SQ_IFETCH[A-Z0-9_]*\|SQ_INST_LEVEL[A-Z0-9_]*\|SQ_WAIT_IFETCH[A-Z0-9_]*" "$OUT/counters.txt" | sort -u > "$OUT/names.txt" || true
cat "$OUT/names.txt"
PASSES=(
 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"
 "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READY SQC_TC_INST_REQ"
)
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 bench.py --steps 4 --warmup 1 --quick --no-cpu-baseline > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed: $P"
  i=$((i+1))
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
grep -A14 "^k_screen_encode" "$OUT/summary.txt"
