#!/bin/bash
# GPU box helper (round 3): GPU suite on the current build, then standard-mode timing A (libmi355jpeg_prev.so) vs B (current)
set -e -o pipefail
OUT=gpurun_out/$1; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
P=$GRAFT_REPO_ROOT/jpeg-encoder-opencl_amd
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1 || { tail -60 "$OUT/pytest_gpu.log"; exit 1; }
tail -2 "$OUT/pytest_gpu.log"
for r in 1 2; do for f in 6 2; do
  MI355_DIAG_FLAGS=$f MI355_JPEG_LIB=$P/libmi355jpeg_prev.so python tools/pipeline_diag.py 128 | sed "s/^/A /" | tee -a "$OUT/std.log"
  MI355_DIAG_FLAGS=$f python tools/pipeline_diag.py 128 | sed "s/^/B /" | tee -a "$OUT/std.log"
done; done
