#!/bin/bash
# GPU box helper: sustained runs (bench.py --steps N, default 8000 = ~35 s each) of several library builds, one after the other, then
# the first one again (does a long run time like a short one, or does the device throttle?).  usage: tools/soak_ab.sh <tag> <lib.so> ...
set -e
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
P=$GRAFT_REPO_ROOT/jpeg-encoder-opencl_amd
i=0
for L in "$@" "$1"; do
  i=$((i+1))
  MI355_JPEG_LIB=$P/$L timeout -k 10 400 python bench.py --quick --no-cpu-baseline --steps ${SOAK_STEPS:-8000} > "$OUT/s$i.json" 2> "$OUT/s$i.err" || { tail -5 "$OUT/s$i.err"; exit 1; }
  python - "$OUT/s$i.json" "$L" <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
print("%-28s %d steps: %.1f Gpx/s  ms/step %.4f  kernel ms/frame %.5f" % (sys.argv[2], j["steps"], j["value"]/1e3, j["ms_per_step"], j["roofline"]["kernel_ms_per_frame"]))
PY
done
