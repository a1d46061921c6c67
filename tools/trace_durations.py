#!/usr/bin/env python3
"""Development helper: mean/max duration per kernel in the multi-stream part of a rocprofv3
--kernel-trace csv (argument: the -d output directory).  Used for DESIGN.md §4.5."""
import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
ks=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0].replace('void mi355::','').replace('mi355::',''),r['Queue_Id']) for r in rows]
ks.sort()
idx=[i for i,k in enumerate(ks) if k[3]=='4']
d=collections.defaultdict(list)
for s,e,n,q in ks[idx[10]:idx[-10]]:
    d[n].append((e-s)/1e3)
print(sys.argv[1], ' '.join('%s %.1f/%.1f' % (n[:14], sum(v)/len(v), max(v)) for n,v in d.items()))
