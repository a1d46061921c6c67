#!/usr/bin/env python3
"""Instruction histogram per kernel of a hipcc -save-temps .s file."""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
cur, cnt = None, {}
for line in txt:
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1)
        cnt[cur] = collections.Counter()
        continue
    if cur and line.startswith("\t.end_amdhsa_kernel"):
        cur = None
    if cur:
        m = re.match(r"\s+([a-z][a-z_0-9]+)(\s|$)", line)
        if m and not m.group(1).startswith("."):
            cnt[cur][m.group(1)] += 1
for k, c in cnt.items():
    tot = sum(c.values())
    if tot == 0:
        continue
    print(k[:70], "total", tot)
    print("   ", c.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 12))
