#!/usr/bin/env python
"""Idle time between consecutive kernels of one stream, from a rocprofv3 kernel trace (csv): for the timed batch's sweep loop
-- level kernel -> control kernel -> level kernel -- how long the queue sits empty at each boundary.
   python tools/trace_gaps.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].split('<')[0].split('::')[-1],
                     int(r.get('Workgroup_Size', 0) or 0), int(r.get('Grid_Size', 0) or 0), r.get('Queue_Id', '')))
rows.sort()
big = max(g for (_, _, n, _, g, _) in rows if n == 'le_level_kernel')
gaps = defaultdict(list)
dur = defaultdict(list)
prev = None
for s, e, n, wg, g, q in rows:
    if prev is not None and (prev[2] in ('le_level_kernel', 'le_control_kernel')) and n in ('le_level_kernel', 'le_control_kernel') and (g == big or prev[4] == big):
        gaps[(prev[2], n)].append((s - prev[1]) / 1e3)
    if n in ('le_level_kernel', 'le_control_kernel') and (g == big or n == 'le_control_kernel'):
        dur[n].append((e - s) / 1e3)
    prev = (s, e, n, wg, g, q)
for k, v in sorted(gaps.items()):
    v = sorted(v)
    print('gap %-18s -> %-18s n=%4d  median %.2f us  mean %.2f us  p90 %.2f us' % (k[0], k[1], len(v), v[len(v) // 2], sum(v) / len(v), v[int(len(v) * 0.9)]))
for k, v in sorted(dur.items()):
    v = sorted(v)
    print('kernel %-18s n=%4d  median %.2f us  mean %.2f us' % (k, len(v), v[len(v) // 2], sum(v) / len(v)))
