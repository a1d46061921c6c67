#!/bin/bash
# GPU box helper: does the wide block-encode kernel lose because its three waves per SIMD (161 registers each) leave the
# tail kernels no registers?  The same kernel at 11, 10 and 9 waves per CU (10: two SIMDs keep 176 registers free) and the default.
set -e -o pipefail
OUT=gpurun_out/$1; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
P=$GRAFT_REPO_ROOT/jpeg-encoder-opencl_amd
for r in 1 2; do
  python bench.py --quick --no-cpu-baseline > "$OUT/classic$r.json" 2> "$OUT/classic$r.err"
  MI355_JPEG_ENCODE_SHAPE=wide python bench.py --quick --no-cpu-baseline > "$OUT/wide11_$r.json" 2> "$OUT/wide11_$r.err"
  MI355_JPEG_ENCODE_SHAPE=wide MI355_JPEG_LIB=$P/libmi355jpeg_w10.so python bench.py --quick --no-cpu-baseline > "$OUT/wide10_$r.json" 2> "$OUT/wide10_$r.err"
  MI355_JPEG_ENCODE_SHAPE=wide MI355_JPEG_LIB=$P/libmi355jpeg_w9.so python bench.py --quick --no-cpu-baseline > "$OUT/wide09_$r.json" 2> "$OUT/wide09_$r.err"
done
python - "$OUT" <<'PY'
import json,sys,glob,os
o=sys.argv[1]
for f in sorted(glob.glob(o+"/*.json")):
    j=json.load(open(f))
    print(os.path.basename(f), "value %.1f Gpx/s  ms/frame %.5f  kernel ms/frame %.5f"%(j["value"]/1e3, j["config"]["ms_per_frame"], j["roofline"]["kernel_ms_per_frame"]))
PY
