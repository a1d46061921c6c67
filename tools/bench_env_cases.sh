#!/bin/bash
# GPU box helper: the driver's bench line (headline + other_configs + standard mode) under several environments, one line per
# case and environment.  usage: tools/bench_env_cases.sh <tag> "<ENV=..>" ...   ("-" = the plain environment)
set -e
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
i=0
for E in "$@"; do
  i=$((i+1))
  if [ "$E" = "-" ]; then EE=""; else EE="$E"; fi
  env $EE python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/b$i.json" 2> "$OUT/b$i.err" || { tail -5 "$OUT/b$i.err"; exit 1; }
  python - "$OUT/b$i.json" "$E" <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
print("%-28s headline %.1f | std420 %.1f | %s" % (sys.argv[2], j["value"]/1e3, j["standard_mode"].get("value",0)/1e3,
      " | ".join("%s %.1f" % (c["case"][:18], c["value"]/1e3) for c in j["other_configs"])))
PY
done
