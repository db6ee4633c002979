#!/usr/bin/env python3
"""Group the le_level_kernel dispatches of a rocprofv3 kernel trace by launch geometry.

A bench.py process launches the kernel for three kinds of plans: the batched units it times, the
one-network units of the latency probe, and set-up runs; `--stats` averages over all of them.  The
geometry (workgroups of the launch) identifies the launches of the timed batch: their average duration
is what bench.py's roofline.us_per_launch must agree with.
usage: tools/rocprof_digest.py <kernel_trace.csv> [bench.json]
"""
import collections
import csv
import json
import sys

all_rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in all_rows if 'le_level_kernel' in r['Kernel_Name']]
groups = collections.OrderedDict()
for r in rows:
    wgs = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])) * int(r['Grid_Size_Y'])
    flat = int(r['Grid_Size_Y']) == 1
    g = groups.setdefault((flat, wgs), [])
    g.append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
print('plan,workgroups,dispatches,avg_us,min_us,max_us')
batched = []
for (flat, wgs), d in sorted(groups.items(), key=lambda kv: (-kv[0][0], kv[0][1])):
    print('%s,%d,%d,%.2f,%.2f,%.2f' % ('batched' if flat else 'single-network', wgs, len(d), sum(d) / len(d) / 1e3,
                                       min(d) / 1e3, max(d) / 1e3))
    if flat:
        batched += d
if len(sys.argv) > 2:
    b = json.loads(open(sys.argv[2]).read().splitlines()[0])
    want = {l['workgroups'] for l in b['roofline']['levels']}
    sel = [x for (flat, wgs), d in groups.items() if flat and wgs in want for x in d]
    print('# launches of the timed batch (%s workgroups): %d dispatches, average %.2f us; bench.py roofline.us_per_launch = %.2f us'
          % (sorted(want), len(sel), sum(sel) / max(len(sel), 1) / 1e3, b['roofline']['us_per_launch']))

# round 6: the lean launches of the free-running layers (one per group of sweeps), by grid size
lean = collections.OrderedDict()
for r in all_rows:
    if 'le_lean_kernel' in r['Kernel_Name']:
        lean.setdefault(int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for wgs, d in sorted(lean.items()):
    print('# le_lean_kernel, %d workgroups: %d dispatches, average %.2f us (min %.2f, max %.2f)' % (wgs, len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3))
