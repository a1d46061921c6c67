#!/bin/bash
# GPU box helper: PMC passes over tools/pipeline_diag.py for the pipeline selected by the environment.
# usage: tools/pmc_diag.sh <outdir> [frames]
set -e
OUT=$1; N=${2:-32}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum"
)
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 tools/pipeline_diag.py $N > "$OUT/pass$i.log" 2>&1
  i=$((i+1))
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
