#!/usr/bin/env python3
"""Developer tool (GPU box): average duration per (kernel, grid size) from a rocprofv3 --kernel-trace CSV -- a bench that runs one kernel at
several problem sizes in one process gets one line per size, which --stats merges.   python tools/kernel_trace_stats.py <dir> [name filter]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ''
f = (glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True) or [None])[0]
if not f:
    sys.exit('no kernel trace under ' + d)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if flt and flt not in n:
        continue
    acc[(n.split('(')[0][-60:], r.get('Grid_Size', r.get('Grid_Size_X', '?')))].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for (n, g), v in sorted(acc.items()):
    v.sort()
    print('%-62s grid %-9s calls %5d  avg %8.2f us  median %8.2f  min %8.2f' % (n, g, len(v), sum(v) / len(v), v[len(v) // 2], v[0]))
