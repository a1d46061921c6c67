#!/usr/bin/env python3
"""profiles/traffic.json from a tools/pmc_run.sh output directory.

usage: tools/make_traffic.py <pmc dir> <summary path recorded as the source> [--sha HEX]

bench.py reports roofline.traffic only while the kernel sources in the tree hash to `kernel_sources_sha` (the sha the
PMC passes were run on; pmc_run.sh writes it to <pmc dir>/kernel_sources_sha on the GPU box).  Units and corrections
follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are KB, collected in
separate passes; on gfx950 FETCH_SIZE tallies the 128-byte read requests of a streaming read at 64 bytes: x2.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FRAMES_PER_LAUNCH = 16  # bench.py: 128 frames per call through the stream pool, 16 frames per launch
W, H = 3840, 2160


def per_dispatch(pmc, counter):
    vals = []
    for path in glob.glob(os.path.join(pmc, "pass*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if "k_screen_encode" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    vals.append(float(row["Counter_Value"]))
    vals.sort()
    return vals


def main():
    pmc, source = sys.argv[1], sys.argv[2]
    sha = None
    if "--sha" in sys.argv:
        sha = sys.argv[sys.argv.index("--sha") + 1]
    elif os.path.exists(os.path.join(pmc, "kernel_sources_sha")):
        sha = open(os.path.join(pmc, "kernel_sources_sha")).read().strip()
    else:
        import bench
        sha = bench.kernel_sources_sha()
    fetch, write, valu = (per_dispatch(pmc, c) for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"))
    assert fetch and write and valu, "missing counters under %s" % pmc
    # the 16-frame launches of the batch (the largest ones; bench.py's second-operating-point gate adds one small launch of
    # another shape): they all have the same shape, the median speaks for all
    fetch, write, valu = ([x for x in v if x > 0.9 * v[-1]] for v in (fetch, write, valu))
    for v in (fetch, write, valu):
        assert v[-1] / v[0] < 1.05, (v[0], v[-1])
    med = lambda v: v[len(v) // 2]
    rd = med(fetch) * 1024 * 2 / FRAMES_PER_LAUNCH
    wr = med(write) * 1024 / FRAMES_PER_LAUNCH
    out = {
        "kernel": "k_screen_encode",
        "kernel_sources_sha": sha,
        "source": source,
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc_run.sh: bench.py --steps 4 --warmup 1 "
                  "--quick, every k_screen_encode launch = %d frames of %dx%d), median over %d launches, divided by the frames "
                  "per launch" % (FRAMES_PER_LAUNCH, W, H, len(fetch)),
        "correction": "FETCH_SIZE x2 on gfx950 (128-byte read requests tallied at 64 bytes, MI355X_MICROARCH.md); check on this "
                      "kernel: the %d RGB bytes of a frame must be fetched at least once and the corrected figure is %.3f of that. "
                      "WRITE_SIZE taken as exact." % (W * H * 3, rd / (W * H * 3)),
        "FETCH_SIZE_KB_median_per_launch": med(fetch),
        "WRITE_SIZE_KB_median_per_launch": med(write),
        "k_screen_encode_hbm_bytes_per_frame": int(rd + wr),
        "breakdown_bytes_per_frame": {"read_corrected": int(rd), "written": int(wr)},
        "k_screen_encode_valu_insts_per_frame": int(med(valu) / FRAMES_PER_LAUNCH),
        "valu_source": "SQ_INSTS_VALU median per launch / frames per launch (wave-level instructions summed over the device)",
    }
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
