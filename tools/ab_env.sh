#!/bin/bash
# GPU box helper: A/B of one library under two environments in ONE call.  usage: tools/ab_env.sh <tag> "<ENV=..>" 
set -e
OUT=gpurun_out/$1; mkdir -p "$OUT"
for r in 1 2; do
  python bench.py --quick --no-cpu-baseline > "$OUT/A$r.json" 2> "$OUT/A$r.err"
  env $2 python bench.py --quick --no-cpu-baseline > "$OUT/B$r.json" 2> "$OUT/B$r.err"
done
python - "$OUT" <<'PY'
import json,sys
o=sys.argv[1]
for k in ("A1","B1","A2","B2"):
    j=json.load(open("%s/%s.json"%(o,k)))
    print(k, "value %.1f Gpx/s  ms/frame %.5f  kernel ms/frame %.5f"%(j["value"]/1e3, j["config"]["ms_per_frame"], j["roofline"]["kernel_ms_per_frame"]))
PY
