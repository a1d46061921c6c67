#!/bin/bash
# GPU box helper: HBM fetch of the block-encode kernel per frame for several launch lengths (frames per step / 8 parts).
# usage: tools/pmc_fetch_vs_launch.sh <outdir> <frames per step> ...
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
for n in "$@"; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/f$n" -- python3 bench.py --steps 3 --warmup 1 --quick --no-cpu-baseline --frames-per-step $n > "$OUT/f$n.log" 2>&1
  python3 - "$OUT/f$n" $n <<'PY'
import csv,glob,sys
vals=[]
for p in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "k_screen_encode" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE": vals.append(float(r["Counter_Value"]))
n=int(sys.argv[2]); vals.sort()
big=[v for v in vals if v > 0.5*vals[-1]]
fpl = n/8.0
print("frames/step %d: %d launches, median FETCH_SIZE %.0f KB -> x2 = %.2f MB per frame (RGB 24.88)"%(n,len(big),big[len(big)//2], big[len(big)//2]*2*1024/fpl/1e6))
PY
done
