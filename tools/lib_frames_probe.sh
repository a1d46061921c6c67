#!/bin/bash
# GPU box helper (round 4): throughput against frames per call (even and uneven parts) for several library builds.
# usage: tools/lib_frames_probe.sh <tag> <lib.so> ...   (paths relative to jpeg-encoder-opencl_amd/)
set -e
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
P=$GRAFT_REPO_ROOT/jpeg-encoder-opencl_amd
for r in 1 2; do
 for F in ${FRAMES:-128 124 132 100 64}; do
  for L in "$@"; do
  MI355_JPEG_LIB=$P/$L python bench.py --quick --no-cpu-baseline --steps 60 --frames-per-step $F > $OUT/x.json 2> $OUT/x.err || { tail -5 $OUT/x.err; exit 1; }
  python - "$L frames=$F" "$OUT/x.json" <<'PY'
import json,sys
j=json.load(open(sys.argv[2]))
print("%-44s value %.1f Gpx/s  ms/frame %.5f  kernel ms/frame %.5f"%(sys.argv[1], j["value"]/1e3, j["config"]["ms_per_frame"], j["roofline"]["kernel_ms_per_frame"]))
PY
  done
 done
done
