#!/usr/bin/env python3
"""Average PMC counter values per kernel from rocprofv3 --pmc csv output."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
            short = name.split("(")[0].replace("void ", "").replace("mi355::", "")
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    if not k.startswith("k_"):
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-28s avg %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
