#!/bin/bash
# GPU box helper (round 3): the whole GPU suite with the wide block-encode kernel, then bench A/B classic vs wide
set -e -o pipefail
OUT=gpurun_out/$1; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
MI355_JPEG_ENCODE_SHAPE=wide timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_wide.log" 2>&1 || { tail -40 "$OUT/pytest_wide.log"; exit 1; }
tail -2 "$OUT/pytest_wide.log"
tools/ab_env.sh $1 MI355_JPEG_ENCODE_SHAPE=wide
