#!/usr/bin/env python
"""Where a position of the one-launch correction chain spends its time (GPU box, a library built with -DDFQ_BC_TRACE=1):
   DFQ_HIP_LIB=$PWD/variants/libdfq_hip_bctrace.so python tools/bc_trace.py [--batch=32] mobilenet_v2
Per chain position (max / min over its workgroups, us since the launch's first stamp): entry, weights quantised, expectation
assembled, matvec done, tail done; and the deltas along the critical path."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from dfq_amd import _ffi

dev = torch.device('cuda', 0)
args = [a for a in sys.argv[1:] if not a.startswith('--batch=')]
batch = max([int(a.split('=')[1]) for a in sys.argv[1:] if a.startswith('--batch=')] + [1])
for net in args or ['mobilenet_v2']:
    unit = bench.make_unit([bench.prepare(net, seed, dev) for seed in range(batch)])
    unit['le'].enqueue(3, restart=True, max_sweeps=3, converge_thres=-1.0, converge_count=10 ** 9)
    unit['le'].query()
    for _ in range(3):
        unit['bc'].run()
        unit['bc'].status()
    buf = np.zeros(32768 * 8, dtype=np.int64)
    n_wg = unit['bc'].chain_workgroups if hasattr(unit['bc'], 'chain_workgroups') else None
    n = _ffi.lib().dfq_bc_debug_trace(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    if n == 0:
        sys.exit('not a trace build (make ... EXTRA=-DDFQ_BC_TRACE=1 and point DFQ_HIP_LIB at it)')
    t = buf.reshape(-1, 8)
    t = t[t[:, 0] != 0]
    t = t[t[:, 0] >= t[:, 0].max() - 100 * 2000]          # (rows of an earlier, larger launch: older than 2 ms)
    step = (t[:, 7] >> 32).astype(int) // batch           # launch-major step table: position x network
    t0 = t[:, 0].min()
    us = (t[:, :5] - t0) / 100.0
    print('# {}: {} workgroups, {} positions; us since the first workgroup entered'.format(net, len(t), step.max() + 1))
    print('# pos  wgs | entry(min..max)  quantised(max) | assembled(min..max)  matvec(max)  tail(max) | wait->assembled  matvec  tail  | since previous tail')
    prev_tail = None
    tot = np.zeros(4)
    for p in range(step.max() + 1):
        m = us[step == p]
        if not len(m):
            continue
        tail = m[:, 4].max()
        gap = (m[:, 2].max() - prev_tail) if prev_tail is not None else float('nan')
        d_mv, d_tail = m[:, 3].max() - m[:, 2].max(), tail - m[:, 3].max()
        inner = ' | factors {:5.2f} sum {:5.2f} barrier {:5.2f}'.format(
            (t[step == p][:, 5].max() - t0) / 100.0 - m[:, 2].max(), (t[step == p][:, 6].max() - t[step == p][:, 5].max()) / 100.0,
            m[:, 3].max() - (t[step == p][:, 6].max() - t0) / 100.0)
        print('{:4d} {:5d} | {:7.2f}..{:7.2f}  {:7.2f} | {:7.2f}..{:7.2f}  {:7.2f}  {:7.2f} | {:6.2f} {:6.2f} {:6.2f} | {:6.2f}'.format(
            p, len(m), m[:, 0].min(), m[:, 0].max(), m[:, 1].max(), m[:, 2].min(), m[:, 2].max(), m[:, 3].max(), tail,
            gap, d_mv, d_tail, (tail - prev_tail) if prev_tail is not None else float('nan')) + inner)
        if prev_tail is not None:
            tot += [gap, d_mv, d_tail, tail - prev_tail]
        prev_tail = tail
    print('# sums over the chain: previous tail -> assembled {:.1f} us, matvec {:.1f}, tail {:.1f}; position to position {:.1f}'.format(*tot))
