#!/bin/bash
# GPU box helper: A/B of two builds of the library in ONE call (boxes differ by a few per cent).
# usage: tools/ab.sh <tag> ; compares libmi355jpeg_prev.so (A) with libmi355jpeg.so (B), two rounds each, interleaved
set -e
OUT=gpurun_out/$1; mkdir -p "$OUT"
P=$GRAFT_REPO_ROOT/jpeg-encoder-opencl_amd
for r in 1 2; do
  MI355_JPEG_LIB=$P/libmi355jpeg_prev.so python bench.py --quick --no-cpu-baseline > "$OUT/A$r.json" 2> "$OUT/A$r.err"
  python bench.py --quick --no-cpu-baseline > "$OUT/B$r.json" 2> "$OUT/B$r.err"
done
python - "$OUT" <<'PY'
import json,sys
o=sys.argv[1]
for k in ("A1","B1","A2","B2"):
    j=json.load(open("%s/%s.json"%(o,k)))
    print(k, "value %.1f Gpx/s  ms/frame %.5f  kernel ms/frame %.5f"%(j["value"]/1e3, j["config"]["ms_per_frame"], j["roofline"]["kernel_ms_per_frame"]))
PY
