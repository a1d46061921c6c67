#!/bin/bash
# GPU box helper (round 3): the GPU suite on the current build, then A/B of libmi355jpeg_prev.so (A) vs current (B)
set -e -o pipefail
OUT=gpurun_out/$1; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1 || { tail -60 "$OUT/pytest_gpu.log"; exit 1; }
tail -2 "$OUT/pytest_gpu.log"
tools/ab.sh $1
