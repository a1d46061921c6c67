#!/usr/bin/env python3
"""Summarises a `rocprofv3 --kernel-trace --stats` run of the driver's bench command
(tools/gpu_round.sh ... prof) next to the JSON line bench.py printed in that same run:

  * per kernel: launches, average / min / max duration;
  * the dominant kernel's launches of the TIMED REGION (those between the two marker launches bench.py puts around it)
    and their average duration, which bench.py's roofline.kernel_ms (HIP events on the launch stream inside the timed
    region, divided by the launches) must agree with.

usage: profile_summary.py <gpurun_out/TAG> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys

tag, out = sys.argv[1], sys.argv[2]
trace = glob.glob(os.path.join(tag, "prof", "*", "*kernel_trace.csv"))[0]
line = json.load(open(os.path.join(tag, "prof_bench.json")))
rows = list(csv.DictReader(open(trace)))
by = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mi355::", "")
    by[name].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["VGPR_Count"]),
                     int(r["LDS_Block_Size"]), int(r["Grid_Size_X"])))
kern = line["roofline"]["kernel"]
dom = sorted(by[[k for k in by if k.startswith(kern + "<false, 0>") or k == kern][0]])
steps, per_step = line["steps"], line["config"]["launches_of_dominant_kernel_per_step"]
n_timed = steps * per_step
# The timed region: the dominant kernel's launches between bench.py's two marker launches (k_lcg_fill on 256 bytes: the only
# launches of that kernel with a grid of one workgroup).  The parts of a batch differ in size since round 4, so a launch's
# duration no longer says which leg it belongs to.
fills = sorted(by.get("k_lcg_fill", []))
small = min(f[4] for f in fills)
marks = [f[0] for f in fills if f[4] == small]
assert len(marks) == 2, "expected the two marker launches of bench.py's timed region, found %d" % len(marks)
timed = [x for x in dom if marks[0] < x[0] < marks[1]]
batch_shape = timed
assert len(timed) == n_timed, (len(timed), n_timed)
avg_us = sum(d for _, d, *_ in timed) / len(timed)
summary = {
    "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --gpus 1 --steps %d --warmup %d" % (line["steps"], line["warmup"]),
    "bench_line": {"value_Mpixel_s": line["value"], "ms_per_step": line["ms_per_step"],
                   "roofline.kernel_ms": line["roofline"]["kernel_ms"], "roofline.frac": line["roofline"]["frac"],
                   "frames_per_launch": line["roofline"]["frames_per_launch"]},
    "dominant_kernel": {"name": kern, "launches_in_trace": len(dom), "launches_of_the_batch_shape": len(batch_shape),
                        "launches_of_the_timed_region": len(timed), "rocprof_avg_us_timed_region": round(avg_us, 2),
                        "rocprof_min_us": round(min(d for _, d, *_ in timed), 2), "rocprof_max_us": round(max(d for _, d, *_ in timed), 2),
                        "bench_events_us": round(line["roofline"]["kernel_ms"] * 1e3, 2),
                        "events_over_rocprof": round(line["roofline"]["kernel_ms"] * 1e3 / avg_us, 4),
                        "vgpr": timed[0][2], "lds_bytes_per_workgroup": timed[0][3], "grid_threads": timed[0][4]},
    "all_kernels": {k: {"launches": len(v), "avg_us": round(sum(d for _, d, *_ in v) / len(v), 2), "min_us": round(min(d for _, d, *_ in v), 2),
                        "max_us": round(max(d for _, d, *_ in v), 2)} for k, v in sorted(by.items()) if k.startswith("k_")},
}
json.dump(summary, open(out, "w"), indent=1)
print(json.dumps(summary["dominant_kernel"]))
