#!/usr/bin/env python3
"""Where a sweep launch's time goes, by KIND of workgroup (tuning aid; run on the GPU box).

usage: tools/trace_classes.py [batch] [net]
Joins dfq_le_trace_blocks (entry / exit of every workgroup of the third sweep) with dfq_le_plan_block_info (what the
workgroup does) and prints, per kind: workgroups, bytes, mean / p50 / p90 residency, share of the launch's summed
residency against share of its bytes, and bytes per workgroup-microsecond.  Then a timeline: per 10 us bucket the
bytes of the workgroups that finished in it and how many were resident."""
import sys
import collections
import torch
sys.path.insert(0, '/root/repo')
sys.path.insert(0, '.')
import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = sys.argv[2] if len(sys.argv) > 2 else 'mobilenet_v2'
dev = torch.device('cuda', 0)
protos = [bench.prepare(net, seed=i, dev=dev) for i in range(batch)]
unit = bench.make_unit(protos)
le = unit['le']
KIND = ['row v4', 'row v1', 'row thread', 'col v4', 'col v1', 'col thread']
for launch in range(le.levels):
    rows = le.trace_blocks(launch)
    info = [le.block_info(launch, b) for b in range(len(rows))]
    t0 = min(r[0] for r in rows if r[1] > 0)
    t1 = max(r[1] for r in rows)
    cls = collections.defaultdict(list)
    for r, i in zip(rows, info):
        if r[1] <= 0:
            continue
        nbytes = 8 * i['rw_elements'] + 4 * i['ro_elements']
        size = 'full' if i['rw_elements'] + i['ro_elements'] >= 6000 else ('half' if i['rw_elements'] + i['ro_elements'] >= 2500 else 'small')
        key = (KIND[i['kind']], 'ro' if i['ro_elements'] else 'rw', 'waits' if i['waits'] else '-', 'pub' if i['publishes'] else '-', size)
        cls[key].append(((r[1] - r[0]) * 10, nbytes, r[0] - t0))
    tot_res = sum(d for v in cls.values() for d, _, _ in v)
    tot_b = sum(b for v in cls.values() for _, b, _ in v)
    print('launch %d: span %.1f us, %d workgroups, %.1f MB, summed residency %.0f us (mean resident %.0f)' % (
        launch, (t1 - t0) / 100.0, len(rows), tot_b / 1e6, tot_res / 1e3, tot_res / ((t1 - t0) * 10.0)))
    print('%-42s %6s %8s %7s %7s %7s %7s %7s %9s' % ('kind', 'wgs', 'MB', 'mean', 'p50', 'p90', 'res %', 'byte %', 'B/wg-us'))
    for key, v in sorted(cls.items(), key=lambda kv: -sum(d for d, _, _ in kv[1])):
        d = sorted(x[0] for x in v)
        b = sum(x[1] for x in v)
        print('%-42s %6d %8.1f %7.0f %7.0f %7.0f %7.1f %7.1f %9.0f' % (
            ' '.join(key), len(v), b / 1e6, sum(d) / len(d), d[len(d) // 2], d[len(d) * 9 // 10],
            100.0 * sum(d) / tot_res, 100.0 * b / tot_b, b / (sum(d) / 1e3)))
    # timeline
    step = 1000     # ticks of 10 ns
    nb = int((t1 - t0) // step) + 1
    done_b = [0] * nb
    res = [0.0] * nb
    for r, i in zip(rows, info):
        if r[1] <= 0:
            continue
        done_b[int((r[1] - t0) // step)] += 8 * i['rw_elements'] + 4 * i['ro_elements']
        a, b = r[0] - t0, r[1] - t0
        for k in range(int(a // step), int(b // step) + 1):
            lo, hi = max(a, k * step), min(b, (k + 1) * step)
            if hi > lo:
                res[k] += (hi - lo) / step
    print('timeline (10 us buckets): TB/s of finished workgroups | mean resident workgroups')
    print('  ' + ' '.join('%.1f' % (x / 10e-6 / 1e12) for x in done_b))
    print('  ' + ' '.join('%d' % x for x in res))
