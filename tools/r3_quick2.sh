#!/bin/bash
# GPU box helper (round 3): parity subset with the tile pipeline forced everywhere + timing of both pipelines
set -e -o pipefail
OUT=gpurun_out/$1; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
MI355_JPEG_PIPELINE=tile timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_standard_mode.py -m gpu -x -q -k "not emit_direct and not both_pipelines and not per_frame and not transform_modes" > "$OUT/pytest_tile.log" 2>&1 || { tail -40 "$OUT/pytest_tile.log"; exit 1; }
tail -2 "$OUT/pytest_tile.log"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "both_pipelines or per_frame" > "$OUT/pytest_both.log" 2>&1 || { tail -40 "$OUT/pytest_both.log"; exit 1; }
tail -2 "$OUT/pytest_both.log"
for p in tile launches tile launches; do
  MI355_JPEG_PIPELINE=$p timeout -k 10 300 python tools/pipeline_diag.py 128 | tee -a "$OUT/diag.log"
done
