#!/usr/bin/env python3
"""Tuning aid (GPU box): convergence-controlled runs of the two sweep kernels, enqueued at once or in chunks."""
import os
import sys
sys.path.insert(0, '/root/repo')
sys.path.insert(0, '.')
import torch
import bench

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device('cuda', 0)


def run(env, chunks, **cfg):
    for k in ('DFQ_LE_PERSIST', 'DFQ_LE_SWEEP_WGS'):
        os.environ.pop(k, None)
    os.environ.update(env)
    protos = [bench.prepare('mobilenet_v2', seed=i, dev=dev) for i in range(batch)]
    unit = bench.make_unit(protos)
    le = unit['le']
    le.enqueue(0, restart=True, **cfg)
    out = []
    for c in chunks:
        le.enqueue(c, restart=False, **cfg)
        res, done = le.query_all()
        out.append([(r['sweeps'], r['done'] if 'done' in r else None) for r in res])
    return out


for env in ({'DFQ_LE_PERSIST': '0'}, {'DFQ_LE_SWEEP_WGS': '1024'}, {'DFQ_LE_SWEEP_WGS': '64'}):
    for chunks in ([80], [8, 16, 32, 32], [40, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1]):
        try:
            print(env, chunks, run(env, chunks)[-1], flush=True)
        except Exception as e:
            print(env, chunks, 'FAILED', str(e)[:90], flush=True)
