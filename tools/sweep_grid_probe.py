#!/usr/bin/env python3
"""Tuning aid (GPU box): run a batched equalisation with le_sweep_kernel at several grid sizes and report which complete."""
import os
import sys
import time
sys.path.insert(0, '/root/repo')
sys.path.insert(0, '.')
import torch
import bench
from dfq_amd import dfq

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device('cuda', 0)
for wgs in sys.argv[2:] or ['64', '256', '512', '768', '1024', '0']:
    if wgs == '0':
        os.environ.pop('DFQ_LE_SWEEP_WGS', None)
    else:
        os.environ['DFQ_LE_SWEEP_WGS'] = wgs
    protos = [bench.prepare('mobilenet_v2', seed=i, dev=dev) for i in range(batch)]
    unit = bench.make_unit(protos)
    le = unit['le']
    t0 = time.time()
    try:
        n = int(os.environ.get('PROBE_SWEEPS', '4'))
        if n > 0:
            le.enqueue(0, restart=True, max_sweeps=n)
            le.enqueue(n, restart=False, max_sweeps=n)
            torch.cuda.synchronize()
        else:
            le.run()
        res, done = le.query_all()
        print('wgs', wgs, 'grid', le.sweep_workgroups, 'tiles', le.level_info(0)['workgroups'], 'ok sweeps', [r['sweeps'] for r in res][:4], '%.3fs' % (time.time() - t0), flush=True)
    except Exception as e:
        print('wgs', wgs, 'grid', le.sweep_workgroups, 'FAILED', str(e)[:100], '%.3fs' % (time.time() - t0), flush=True)
