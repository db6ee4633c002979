#!/usr/bin/env python3
"""Tuning aid (GPU box): fixed number of sweeps with the per-tile kernel and with le_sweep_kernel at several grid sizes;
reports which layers differ."""
import os
import sys
sys.path.insert(0, '/root/repo')
sys.path.insert(0, '.')
import torch
import torch.nn as nn
import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device('cuda', 0)


def run(env):
    for k in ('DFQ_LE_PERSIST', 'DFQ_LE_SWEEP_WGS'):
        os.environ.pop(k, None)
    os.environ.update(env)
    protos = [bench.prepare('mobilenet_v2', seed=i, dev=dev) for i in range(batch)]
    unit = bench.make_unit(protos)
    le = unit['le']
    cfg = dict(max_sweeps=n) if os.environ.get('PROBE_CONVERGE') else dict(max_sweeps=n, converge_thres=-1.0, converge_count=10 ** 9)
    le.enqueue(0, restart=True, **cfg)
    le.enqueue(n, restart=False, **cfg)
    res, done = le.query_all()
    le.stage.writeback()
    ws = []
    for ni, (model, graph, bottoms, rels) in enumerate(unit['nets']):
        for k in graph:
            m = graph[k]
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                ws.append(('net%d.%s' % (ni, k), tuple(m.weight.shape), m.weight.detach().clone()))
    return ws, [r.get('diff') for r in res], le.sweep_workgroups


ref, dref, _ = run({'DFQ_LE_PERSIST': '0'})
for wgs in sys.argv[3:] or ['1024', '64']:
    got, dgot, grid = run({'DFQ_LE_SWEEP_WGS': wgs})
    bad = [(i, k, shp, int((a != b).sum().item())) for i, ((k, shp, a), (_, _, b)) in enumerate(zip(ref, got)) if not torch.equal(a, b)]
    print('wgs', wgs, 'grid', grid, 'layers', len(ref), 'differing', len(bad), 'diff', dref[:3], dgot[:3])
    for b in bad[:40]:
        print('   ', b)
