#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/$1; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
P=$GRAFT_REPO_ROOT/jpeg-encoder-opencl_amd
for r in 1 2; do
  MI355_DIAG_CAP_MB=16 MI355_DIAG_FLAGS=2 MI355_JPEG_LIB=$P/libmi355jpeg_prev.so python tools/pipeline_diag.py 64 | sed "s/^/A /" | tee -a "$OUT/std.log"
  MI355_DIAG_CAP_MB=16 MI355_DIAG_FLAGS=2 python tools/pipeline_diag.py 64 | sed "s/^/B /" | tee -a "$OUT/std.log"
done
