#!/usr/bin/env python
"""Single-network latency probe (GPU box): LE / BC GPU time and wall time of one pass, per network.
   python tools/lat.py mobilenet_v2 deeplab_mnv2:60 resnet18     (environment switches of the library apply)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

dev = torch.device('cuda', 0)
for item in sys.argv[1:] or ['mobilenet_v2']:
    net, _, pin = item.partition(':')
    proto = bench.prepare(net, 0, dev)
    if pin:
        sweeps, pinned = int(pin), True
    else:
        sweeps, pinned = bench.make_unit([proto])['le'].run()['sweeps'], False
    best = None
    for _ in range(3):
        m = bench.single_network_pass(proto, sweeps, pinned=pinned)
        if best is None or m['pass_ms'] < best['pass_ms']:
            best = m
    print(json.dumps({'net': net, 'sweeps': sweeps, 'pass_ms': round(best['pass_ms'], 4), 'le_ms': round(best['equalization_ms'], 4),
                      'bc_ms': round(best['bias_correction_ms'], 4), 'tiles': best['resident_tiles'],
                      'us_per_sweep': round(best['equalization_ms'] * 1e3 / max(1, sweeps), 2)}))
