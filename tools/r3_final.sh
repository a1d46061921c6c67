#!/bin/bash
# GPU box helper (round 3): the round's profile bundle in one call -- GPU suite, driver bench command, rocprofv3
# kernel trace + stats of the same command, config bench, PMC passes (strict default; standard 4:2:0 leg separately).
set -e -o pipefail
TAG=$1
tools/gpu_round.sh $TAG test bench prof cfg
tools/pmc_run.sh gpurun_out/$TAG/pmc > /dev/null
# standard mode (4:2:0) in passes of its own
OUT=gpurun_out/$TAG/pmc_std; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for P in "${PASSES[@]}"; do
  MI355_DIAG_FLAGS=6 rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 tools/pipeline_diag.py 64 > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed"
  i=$((i+1))
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
find gpurun_out/$TAG -name "*.csv" -size +2M -delete
echo done
