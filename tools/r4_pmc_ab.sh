#!/bin/bash
# GPU box helper (round 4): instruction-count and wait counters of the block-encode kernel for several library builds.
# usage: tools/r4_pmc_ab.sh <tag> <lib.so> [<lib.so> ...]   (paths relative to jpeg-encoder-opencl_amd/)
# One rocprofv3 process per counter group and library (counters only, no tracing domains); the program directly after `--`.
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# equal parts (16 frames per launch): the summaries divide a launch's counters by its frames
export MI355_JPEG_TAPER=0
P=$GRAFT_REPO_ROOT/jpeg-encoder-opencl_amd
PASSES=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS"
)
for L in "$@"; do
  OUT=gpurun_out/$TAG/$L; mkdir -p "$OUT"
  export MI355_JPEG_LIB=$P/$L
  i=0
  for C in "${PASSES[@]}"; do
    rocprofv3 --pmc $C --output-format csv -d "$OUT/pass$i" -- python3 bench.py --steps 4 --warmup 1 --quick --no-cpu-baseline > "$OUT/pass$i.log" 2>&1
    i=$((i+1))
  done
  python3 tools/pmc_summary.py "$OUT" > "$OUT/summary.txt"
  echo "== $L"; grep -A17 "^k_screen_encode" "$OUT/summary.txt" | head -18
  find "$OUT" -name "*.csv" -size +5M -delete
done
