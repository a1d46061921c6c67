#!/bin/bash
# GPU box helper: PMC counter passes over a short bench run (one rocprofv3 process per pass;
# counters only, no tracing domains).  Usage: tools/pmc_run.sh <outdir> [bench args...]
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# equal parts (16 frames per launch): the summaries divide a launch's counters by its frames
export MI355_JPEG_TAPER=0
mkdir -p "$OUT"
python3 -c "import bench; print(bench.kernel_sources_sha())" > "$OUT/kernel_sources_sha"
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY"
 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_IFETCH"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "GRBM_GUI_ACTIVE"
 "TCC_HIT_sum TCC_MISS_sum"
)
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 bench.py --steps 4 --warmup 1 --quick --no-cpu-baseline "$@" > "$OUT/pass$i.log" 2>&1
  i=$((i+1))
done
python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
