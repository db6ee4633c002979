#!/usr/bin/env python3
"""Tuning aid (GPU box): what the HOST spends per batch -- building the two plans (profiled) and enqueueing
one step -- next to the GPU time of that step.   usage: tools/host_cost.py [batch]"""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, '.')
import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda', 0)
protos = [bench.prepare('mobilenet_v2', seed=i, dev=dev) for i in range(batch)]
units = [bench.make_unit(protos) for _ in range(6)]
print('plan build per batch of %d: %.1f ms' % (batch, sum(u['plan_build_ms'] for u in units) / len(units)))
probe = bench.make_unit(protos)
probe['le'].run()
sweeps = max(r['sweeps'] for r in probe['le'].query_all()[0])


def step(u):
    u['le'].enqueue(sweeps, restart=True, max_sweeps=sweeps)
    u['bc'].run()


for u in units[:2]:
    step(u)
torch.cuda.synchronize()
t0 = time.perf_counter()
for u in units[2:]:
    step(u)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
n = len(units) - 2
print('host enqueue per step %.3f ms (%d launches), GPU-bound total per step %.3f ms'
      % ((t1 - t0) * 1e3 / n, sweeps * (probe['le'].levels + 1) + 60, (t2 - t0) * 1e3 / n))
pr = cProfile.Profile()
pr.enable()
bench.make_unit(protos)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
