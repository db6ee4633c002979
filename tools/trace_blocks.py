#!/usr/bin/env python3
"""Timeline of the workgroups of every equalisation launch (tuning aid; run on the GPU box).

usage: tools/trace_blocks.py [batch] [net]
Prints per launch: workgroups, span of the launch, mean/percentile workgroup duration, how many
workgroups were resident at once (chip and per CU), and the time the first / last workgroup started.
"""
import sys
import collections
import torch
sys.path.insert(0, '/root/repo')
sys.path.insert(0, '.')
import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
net = sys.argv[2] if len(sys.argv) > 2 else 'mobilenet_v2'
dev = torch.device('cuda', 0)
protos = [bench.prepare(net, seed=i, dev=dev) for i in range(batch)]
unit = bench.make_unit(protos)
le = unit['le']
for launch in range(le.levels):
    info = le.level_info(launch)
    rows = [r for r in le.trace_blocks(launch) if r[1] > 0]
    t0 = min(r[0] for r in rows)
    t1 = max(r[1] for r in rows)
    dur = sorted((r[1] - r[0]) * 10 for r in rows)                # ns (100 MHz clock)
    ev = sorted([(r[0], 1) for r in rows] + [(r[1], -1) for r in rows])
    cur = peak = 0
    area = 0
    last = t0
    for t, d in ev:
        area += cur * (t - last)
        last = t
        cur += d
        peak = max(peak, cur)
    cus = collections.Counter((r[2] >> 32, (r[2] >> 8) & 0xf, (r[2] >> 13) & 0x7) for r in rows)   # xcc, cu, se (HW_ID layout)
    starts = sorted((r[0] - t0) * 10 for r in rows)
    print('launch %d: %d workgroups, %d bytes | span %.2f us | wg ns mean %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f | '
          'resident mean %.0f peak %d | distinct (xcc,cu,se) %d, wgs per CU min %d max %d | last wg starts at %.2f us' % (
              launch, len(rows), 8 * info['rw_elements'] + 4 * info['ro_elements'], (t1 - t0) / 100.0,
              sum(dur) / len(dur), dur[len(dur) // 10], dur[len(dur) // 2], dur[len(dur) * 9 // 10], dur[-1],
              area / max(t1 - t0, 1), peak, len(cus), min(cus.values()), max(cus.values()), starts[-1] / 1e3))
    # start-time histogram in 1 us buckets
    hist = collections.Counter(int(s // 1000) for s in starts)
    print('   starts per us:', ' '.join('%d' % hist.get(i, 0) for i in range(int(starts[-1] // 1000) + 1)))
