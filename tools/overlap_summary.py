#!/usr/bin/env python3
"""Summary of a `rocprofv3 --kernel-trace` run of tools/overlap_probe.py: k_screen_encode alone (16-frame calls) against
k_screen_encode beside the previous part's tail kernels (parts 2..8 of 128-frame calls), and k_merge in both situations.
usage: overlap_summary.py <trace dir> <out.json>"""
import csv
import glob
import json
import os
import sys

trace = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
enc = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_screen_encode" in r["Kernel_Name"]]
mrg = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_merge" in r["Kernel_Name"]]
assert len(enc) == 12 + 5 * 8, len(enc)
dur = lambda x: (x[1] - x[0]) / 1e3
alone = [dur(x) for x in enc[2:12]]
batch = enc[12 + 8:]                                  # four timed 128-frame calls (the first one warms up)
beside = [dur(x) for i, x in enumerate(batch) if i % 8 != 0]
first = [dur(x) for i, x in enumerate(batch) if i % 8 == 0]
m_alone = [dur(x) for x in mrg[2:12]]
m_batch = mrg[12 + 8:]
m_beside = [dur(x) for i, x in enumerate(m_batch) if i % 8 != 7]   # the last part's merge runs after the last encode
m_last = [dur(x) for i, x in enumerate(m_batch) if i % 8 == 7]
avg = lambda v: round(sum(v) / len(v), 2)
out = {"k_screen_encode_us_per_16_frame_launch": {"alone (16-frame calls, one part)": avg(alone), "beside the tails of the part in front (parts 2..8 of 128-frame calls)": avg(beside),
                                                  "first part of a 128-frame call": avg(first), "min/max beside": [round(min(beside), 2), round(max(beside), 2)]},
       "k_merge_us_per_16_frames": {"alone (after its encode, 16-frame calls)": avg(m_alone), "beside the next part's encode": avg(m_beside),
                                    "last part of a 128-frame call (nothing beside it)": avg(m_last), "max beside": round(max(m_beside), 2)},
       "cost_of_the_overlap_for_the_encode_kernel": "%.1f %%" % (100.0 * (avg(beside) / avg(alone) - 1.0)),
       "k_merge_max_over_encode_avg": round(max(m_beside) / avg(beside), 3)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
