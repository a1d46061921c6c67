#!/bin/bash
# GPU box helper: N-way comparison of library builds in ONE call (boxes differ by a few per cent).
# usage: tools/abn.sh <tag> <lib1.so> <lib2.so> ...   (paths relative to jpeg-encoder-opencl_amd/)
# Every library runs the verified quick bench twice (interleaved) and the tools/config_bench.py cases once.
set -e
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
P=$GRAFT_REPO_ROOT/jpeg-encoder-opencl_amd
for r in $(seq 1 ${ABN_ROUNDS:-2}); do
  for L in "$@"; do
    MI355_JPEG_LIB=$P/$L python bench.py --quick --no-cpu-baseline --steps ${ABN_STEPS:-20} > "$OUT/$L.$r.json" 2> "$OUT/$L.$r.err"
    python - "$OUT/$L.$r.json" "$L" <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
print("%-28s value %.1f Gpx/s  ms/frame %.5f  kernel ms/frame %.5f"%(sys.argv[2], j["value"]/1e3, j["config"]["ms_per_frame"], j["roofline"]["kernel_ms_per_frame"]))
PY
  done
done
[ -n "$ABN_NO_CFG" ] && exit 0
for L in "$@"; do
  echo "$L"; MI355_JPEG_LIB=$P/$L python tools/config_bench.py 2>/dev/null | grep "^{" | python -c "import sys,json; [print('  ', j['case'], j['Gpixel_per_s'], j['stage_ms']) for j in map(json.loads, sys.stdin)]"
done
