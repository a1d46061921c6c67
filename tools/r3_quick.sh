#!/bin/bash
# GPU box helper (round 3): parity of the pipelines (one test) + timing of both + phase stamps of the tile kernel
set -e -o pipefail
OUT=gpurun_out/$1; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "both_pipelines or golden or fruit or stage_probes or config" > "$OUT/pytest_gpu.log" 2>&1 || { tail -60 "$OUT/pytest_gpu.log"; exit 1; }
tail -3 "$OUT/pytest_gpu.log"
for p in tile launches tile launches; do
  MI355_JPEG_PIPELINE=$p timeout -k 10 300 python tools/pipeline_diag.py 128 | tee -a "$OUT/diag.log"
done
tools/r3_stamps.sh $1
