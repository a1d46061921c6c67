#!/bin/bash
# GPU box helper (round 4): the GPU suite on the current build, then an N-way A/B of library builds in the same call.
# usage: tools/r4_ab.sh <tag> <lib.so> [<lib.so> ...]   (paths relative to jpeg-encoder-opencl_amd/; the suite runs on libmi355jpeg.so)
set -e -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1 || { tail -60 "$OUT/pytest_gpu.log"; exit 1; }
tail -2 "$OUT/pytest_gpu.log"
tools/abn.sh "$TAG" "$@"
