#!/usr/bin/env python3
"""Development helper: the last N launches of k_screen_encode and the tail kernels of a rocprofv3 --kernel-trace csv
(argument: the -d output directory [N]) as a timeline: start (us from the first shown), duration, queue, grid."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(csv.DictReader(open(f)))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void mi355::', '').replace('mi355::', ''),
             r['Queue_Id'], r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Grid_Size_Y', '')) for r in rows)
ks = [k for k in ks if k[2].startswith(('k_screen_encode', 'k_merge', 'k_dc_heads', 'k_tile_scan'))][-n:]
t0 = ks[0][0]
for s, e, name, q, gx, gy in ks:
    print("%9.1f  %8.1f us  q%-3s %-28s grid %s x %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name[:28], gx, gy))
