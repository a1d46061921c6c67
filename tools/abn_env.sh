#!/bin/bash
# GPU box helper: N-way comparison of ONE library under several environments in one call (boxes differ by a few per cent).
# usage: tools/abn_env.sh <tag> "<ENV=..>" "<ENV=..>" ...   ("-" = the plain environment); ABN_ROUNDS / ABN_STEPS as in abn.sh
set -e
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
for r in $(seq 1 ${ABN_ROUNDS:-3}); do
  i=0
  for E in "$@"; do
    i=$((i+1))
    if [ "$E" = "-" ]; then python bench.py --quick --no-cpu-baseline --steps ${ABN_STEPS:-100} > "$OUT/e$i.$r.json" 2> "$OUT/e$i.$r.err"
    else env $E python bench.py --quick --no-cpu-baseline --steps ${ABN_STEPS:-100} > "$OUT/e$i.$r.json" 2> "$OUT/e$i.$r.err"; fi
    python - "$OUT/e$i.$r.json" "$E" <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
print("%-36s value %.1f Gpx/s  ms/frame %.5f  kernel ms/frame %.5f"%(sys.argv[2], j["value"]/1e3, j["config"]["ms_per_frame"], j["roofline"]["kernel_ms_per_frame"]))
PY
  done
done
