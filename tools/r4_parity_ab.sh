#!/bin/bash
# GPU box helper (round 4): for each library build, a parity subset (golden vectors, stage probes, random sweep, standard
# mode) with that build loaded, then the N-way timing of tools/abn.sh.
# usage: tools/r4_parity_ab.sh <tag> <lib.so> [<lib.so> ...]   (paths relative to jpeg-encoder-opencl_amd/)
set -e -o pipefail
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
P=$GRAFT_REPO_ROOT/jpeg-encoder-opencl_amd
for L in "$@"; do
  [ "$L" = libmi355jpeg_base.so ] && continue
  MI355_JPEG_LIB=$P/$L timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_standard_mode.py -m gpu -x -q \
      -k "golden or probes or random or full_size or equals_checker or boundaries or sweep" > "$OUT/parity_$L.log" 2>&1 \
      || { echo "PARITY FAILED for $L"; tail -30 "$OUT/parity_$L.log"; exit 1; }
  echo "$L: $(tail -1 "$OUT/parity_$L.log")"
done
tools/abn.sh "$TAG" "$@"
