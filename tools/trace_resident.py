#!/usr/bin/env python
"""Where a sweep of the persistent equalisation launch spends its time: per layer, the phase durations of its tiles
(dfq_le_resident_trace).   python tools/trace_resident.py [net] [sweeps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn as nn

from dfq_amd import dfq, synthetic
from dfq_amd.utils import layer_transform as lt
from dfq_amd.utils import relation as rel

TARG = [nn.Conv2d, nn.Linear]
net = sys.argv[1] if len(sys.argv) > 1 else 'mobilenet_v2'
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device('cuda', 0)
model, graph, bottoms = synthetic.build(net, seed=0)
model.to(dev)
lt.merge_batchnorm(model, graph, bottoms, TARG)
rels = rel.create_relation(graph, bottoms, TARG)
plan = dfq.build_le_plan(graph, rels, TARG)
print('tiles', plan.resident_tiles, plan.resident_reason)
plan.resident_trace(sweeps)                      # warm
tiles = plan.resident_trace(sweeps)
t0 = min(t['stamps'][0][0] for t in tiles if t['stamps'][0][0])
names = ['start', 'sA', 'rowpub', 'sB', 'newvals', 'partial', 'end', 'p3a', 'p3b', 'p3c', 'end2']
by_layer = {}
for t in tiles:
    by_layer.setdefault(t['layer'], []).append(t)
for k in (1, 3, 5):
    print('---- sweep', k, '(us since launch start; per layer: max over its tiles) ----')
    for layer in sorted(by_layer):
        ts = by_layer[layer]
        row = []
        for p in range(11):   # the first eleven slots: the classic points
            vals = [t['stamps'][k][p] for t in ts if t['stamps'][k][p]]
            row.append((max(vals) - t0) / 100.0 if vals else float('nan'))
        print('layer {:3d} tiles {:3d} [{:4d}x{:4d}] '.format(layer, len(ts), ts[0]['rows'], ts[0]['cols']) +
              ' '.join('{}={:7.2f}'.format(n, v) for n, v in zip(names, row)))
# per-tile phase DURATIONS (median / max over the layer's tiles), sweeps 2..5 pooled: where a tile's own time goes
import statistics
# stamp slots in program order (index = point, minus one past 7): start waited sA pre-rowstats rowstats published rowpub sB p3a p3b p3c newvals partial end
order = [0, 14, 1, 11, 12, 13, 2, 3, 7, 8, 9, 4, 5, 10]
labels = ['wait', 'read+solveA', 'sync', 'rowstats', 'sync+publish', 'arrive', 'wait+solveB', 'zero', 'p3 compute+commit', 'sync', 'pub cols', 'partial', 'o-vec']
print('---- per-tile phase durations, us (median | max over tiles and sweeps 2-5) ----')
for layer in sorted(by_layer):
    ts = by_layer[layer]
    cells = []
    for a, b, lab in zip(order[:-1], order[1:], labels):
        d = []
        for t in ts:
            for k in (2, 3, 4, 5):
                x, y = t['stamps'][k][a], t['stamps'][k][b]
                if a == 0 and b == 1 and not y:        # chain start: no phase 1
                    continue
                if x and y:
                    d.append((y - x) / 100.0)
        cells.append('{}={:5.2f}|{:5.2f}'.format(lab, statistics.median(d), max(d)) if d else '{}=  -  '.format(lab))
    tot = [(t['stamps'][k + 1][0] - t['stamps'][k][0]) / 100.0 for t in ts for k in (2, 3, 4) if t['stamps'][k][0] and t['stamps'][k + 1][0]]
    print('layer {:3d} x{:3d} [{:4d}x{:4d}] sweep={:5.2f} '.format(layer, len(ts), ts[0]['rows'], ts[0]['cols'], statistics.median(tot) if tot else 0.0) + ' '.join(cells))
dec = sorted(max(t['stamps'][k][6] for t in tiles) for k in range(6))
print('sweep boundaries (us):', [round((d - t0) / 100.0, 2) for d in dec])
print('rollbacks of the traced launch:', plan.resident_stats())
