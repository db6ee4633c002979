#!/usr/bin/env python
"""The last N dispatches of a rocprofv3 kernel trace: name, workgroups, VGPRs, start (us since the first of them), duration, gap.
   python tools/trace_tail.py <kernel_trace.csv> [N]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'dfq::' in r['Kernel_Name']][-int(sys.argv[2]) if len(sys.argv) > 2 else -16:]
t0 = int(rows[0]['Start_Timestamp'])
prev = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    wg = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_X']))
    print('%-36s wg %6d vgpr %3s lds %6s  start %8.1f  dur %7.1f  gap %6.1f' % (
        r['Kernel_Name'].split('(')[0].replace('void ', '')[:36], wg, r['VGPR_Count'], r['LDS_Block_Size'], (s - t0) / 1e3, (e - s) / 1e3,
        (s - prev) / 1e3 if prev else 0.0))
    prev = e
