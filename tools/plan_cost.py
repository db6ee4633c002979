#!/usr/bin/env python
"""Where building the two plans of a batch goes (GPU box): table building in Python vs the C-side plan creation.
   python tools/plan_cost.py [batch]"""
import copy
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn as nn

import bench
from dfq_amd import _ffi, arena, dfq

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device('cuda', 0)
protos = [bench.prepare('mobilenet_v2', seed=i, dev=dev) for i in range(batch)]
for rep in range(5):
    nets = [copy.deepcopy(p) for p in protos]
    torch.cuda.synchronize()
    stage = _ffi.Stage()
    t0 = time.perf_counter()
    lt = dfq._fast_le_tables([(g, r) for (_, g, _, r) in nets], bench.TARG, stage.device)
    t1 = time.perf_counter()
    le = dfq.LEPlan(lt, None, stage=stage)
    t2 = time.perf_counter()
    bt = dfq._fast_bc_tables([(g, b) for (_, g, b, _) in nets], bench.TARG, nn.BatchNorm2d, stage.device)
    t3 = time.perf_counter()
    bc = dfq.BCPlan(bt, None, stage=stage)
    t4 = time.perf_counter()
    print('LE tables %.2f ms, LE plan (C) %.2f ms, BC tables %.2f ms, BC plan (C) %.2f ms, total %.2f ms' %
          ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3))
    le.close(); bc.close()
    del le, bc, lt, bt, nets          # the previous repetition's 32 models go NOW, not inside the next repetition's timed lines
    gc.collect()

# the same batch as ONE allocation (dfq_amd/arena.py): layout once, then plans from one network's tables + base addresses
for rep in range(5):
    nets = [copy.deepcopy(p) for p in protos]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nb = arena.NetworkBatch([(g, b, r) for (_, g, b, r) in nets], bench.TARG)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    le = nb.le_plan()
    t2 = time.perf_counter()
    bc = nb.bc_plan()
    t3 = time.perf_counter()
    print('one allocation: layout %.2f ms (once per batch), LE plan %.2f ms, BC plan %.2f ms, plans %.2f ms' %
          ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t1) * 1e3))
    le.close(); bc.close()
    del le, bc, nb, nets
    gc.collect()
