#!/bin/bash
# GPU box helper: timing-only A/B for diagnostic builds whose output is wrong on purpose (tools/config_bench.py does
# not verify).  A = libmi355jpeg_prev.so, B = current.
set -e
P=$GRAFT_REPO_ROOT/jpeg-encoder-opencl_amd
for r in 1 2; do
  echo "A$r"; MI355_JPEG_LIB=$P/libmi355jpeg_prev.so python tools/config_bench.py "$@" 2>/dev/null | python -c "import sys,json; [print('  ', j['case'], j['Gpixel_per_s'], j['stage_ms']['transform_ms']) for j in map(json.loads, sys.stdin)]"
  echo "B$r"; python tools/config_bench.py "$@" 2>/dev/null | python -c "import sys,json; [print('  ', j['case'], j['Gpixel_per_s'], j['stage_ms']['transform_ms']) for j in map(json.loads, sys.stdin)]"
done
