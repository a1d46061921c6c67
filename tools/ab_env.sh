#!/bin/bash
# GPU box helper: the quick bench under several settings of one environment variable, interleaved twice.
# usage: tools/ab_env.sh <tag> <VAR> <value> [<value> ...]   ("-" = unset)
set -e
OUT=gpurun_out/$1; VAR=$2; shift 2; mkdir -p "$OUT"
for r in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then python bench.py --quick --no-cpu-baseline > "$OUT/${VAR}_unset_$r.json" 2> "$OUT/err.txt"
    else env $VAR=$v python bench.py --quick --no-cpu-baseline > "$OUT/${VAR}_${v}_$r.json" 2> "$OUT/err.txt"; fi
  done
done
python - "$OUT" <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    j=json.load(open(f))
    print(os.path.basename(f), "value %.1f Gpx/s  kernel ms/frame %.5f"%(j["value"]/1e3, j["roofline"]["kernel_ms_per_frame"]))
PY
