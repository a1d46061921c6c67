#!/bin/bash
# GPU box helper: k_merge at wave priority 3 (default) against 1 and 0 (it then takes only issue slots the block-encode
# kernel of the next part leaves idle): bench --quick, interleaved.  The two libraries are builds with the s_setprio(3)
# of k_merge (jpeg_screen_kernels.hip) edited to 1 and 0; that edit is not in the tree.
set -e -o pipefail
OUT=gpurun_out/$1; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
P=$GRAFT_REPO_ROOT/jpeg-encoder-opencl_amd
for r in 1 2; do
  python bench.py --quick --no-cpu-baseline > "$OUT/prio3_$r.json" 2> "$OUT/prio3_$r.err"
  MI355_JPEG_LIB=$P/libmi355jpeg_mp1.so python bench.py --quick --no-cpu-baseline > "$OUT/prio1_$r.json" 2> "$OUT/prio1_$r.err"
  MI355_JPEG_LIB=$P/libmi355jpeg_mp0.so python bench.py --quick --no-cpu-baseline > "$OUT/prio0_$r.json" 2> "$OUT/prio0_$r.err"
done
python - "$OUT" <<'PY'
import json,sys,glob,os
o=sys.argv[1]
for f in sorted(glob.glob(o+"/*.json")):
    j=json.load(open(f))
    print(os.path.basename(f), "value %.1f Gpx/s  ms/frame %.5f  kernel ms/frame %.5f"%(j["value"]/1e3, j["config"]["ms_per_frame"], j["roofline"]["kernel_ms_per_frame"]))
PY
