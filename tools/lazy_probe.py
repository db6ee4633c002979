#!/usr/bin/env python
"""One batch of networks through the lazy-scale engine (for rocprofv3 --kernel-trace --stats).  python tools/lazy_probe.py [batch] [reps]"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from dfq_amd import dfq
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device('cuda', 0)
protos = [bench.prepare('mobilenet_v2', i, dev) for i in range(batch)]
units = []
for _ in range(reps):
    nets = [copy.deepcopy(p) for p in protos]
    units.append(dfq.LazyLEPlan([(g, r) for (_, g, _, r) in nets], bench.TARG))
for u in units:
    ms = bench._gpu_elapsed_ms(lambda: u.run(47))
    print('lazy LE ms', ms)
