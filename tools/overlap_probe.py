#!/usr/bin/env python
"""Does memory-bound work on a SECOND (low-priority) stream fit into the launch gaps of the batch's equalisation loop?
   T(loop alone), T(background alone), T(both): if T(both) - T(loop) << T(background) the gaps take it.
   python tools/overlap_probe.py [batch] [background launches per loop] [MB moved per background launch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n_bg = int(sys.argv[2]) if len(sys.argv) > 2 else 6
mb = float(sys.argv[3]) if len(sys.argv) > 3 else 334.0
dev = torch.device('cuda', 0)
protos = [bench.prepare('mobilenet_v2', i, dev) for i in range(batch)]
probe = bench.make_unit(protos)
probe['le'].run()
sweeps = max(r['sweeps'] for r in probe['le'].query_all()[0])
units = [bench.make_unit(protos) for _ in range(10)]
x = torch.ones(int(mb * 1e6 / 8), device=dev)           # mul_: 4 B read + 4 B written per element
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (0, -1)
sa = torch.cuda.Stream(dev, priority=-1)
sb = torch.cuda.Stream(dev, priority=0)


def loop(u):
    with torch.cuda.stream(sa):
        u['le'].enqueue(sweeps, restart=True, max_sweeps=sweeps)


def background():
    with torch.cuda.stream(sb):
        for _ in range(n_bg):
            x.mul_(1.0)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


loop(units[0]); background(); torch.cuda.synchronize()
for rnd in range(3):
    a = timed(lambda: loop(units[1 + 3 * rnd]))
    b = timed(background)
    both = timed(lambda: (background(), loop(units[2 + 3 * rnd])))
    both2 = timed(lambda: (loop(units[3 + 3 * rnd]), background()))
    print('round {}: loop {:.3f} ms  background {:.3f} ms ({} x {:.0f} MB)  both {:.3f} / {:.3f} ms  -> extra {:.3f} / {:.3f} of {:.3f}'.format(
        rnd, a, b, n_bg, mb, both, both2, both - a, both2 - a, b))
