#!/usr/bin/env python
"""Where a drop-in call on a CPU-resident model spends its time (GPU box): staging (pack, H2D), plan, GPU pass, write-back
(D2H, unpack).   python tools/pcie_cost.py [net] [reps]"""
import contextlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from dfq_amd import _ffi, dfq, synthetic
from dfq_amd.utils import layer_transform as lt
from dfq_amd.utils import relation as rel

net = sys.argv[1] if len(sys.argv) > 1 else 'mobilenet_v2'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
spans = {}


def timed(name, fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            spans[name] = spans.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    return wrapper


_ffi.Stage.prefetch = timed('prefetch', _ffi.Stage.prefetch)
_ffi.Stage.writeback = timed('writeback', _ffi.Stage.writeback)
_ffi.Stage.out_like_many = timed('out_like_many', _ffi.Stage.out_like_many)
dfq.build_le_plan = timed('le_plan', dfq.build_le_plan)
dfq.build_bc_plan = timed('bc_plan', dfq.build_bc_plan)
dfq.LEPlan.run = timed('le_run', dfq.LEPlan.run)
for i in range(reps):
    model, graph, bottoms = synthetic.build(net, seed=0)
    with contextlib.redirect_stdout(sys.stderr):
        lt.merge_batchnorm(model, graph, bottoms, bench.TARG)
        rels = rel.create_relation(graph, bottoms, bench.TARG, delete_single=False)
        spans.clear()
        t0 = time.perf_counter()
        dfq.cross_layer_equalization(graph, rels, bench.TARG)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        le_spans = dict(spans)
        spans.clear()
        dfq.bias_correction(graph, bottoms, bench.TARG)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print('rep %d: LE %.2f ms %s | BC %.2f ms %s | total %.2f ms | plan cache %s' % (
        i, (t1 - t0) * 1e3, {k: round(v, 2) for k, v in le_spans.items()}, (t2 - t1) * 1e3,
        {k: round(v, 2) for k, v in spans.items()}, (t2 - t0) * 1e3, dfq.plan_cache_stats))

import dfq_amd
for i in range(reps):
    model, graph, bottoms = synthetic.build(net, seed=0)
    with contextlib.redirect_stdout(sys.stderr):
        lt.merge_batchnorm(model, graph, bottoms, bench.TARG)
        rels = rel.create_relation(graph, bottoms, bench.TARG, delete_single=False)
        spans.clear()
        t0 = time.perf_counter()
        with dfq_amd.staging() as st:
            dfq.cross_layer_equalization(graph, rels, bench.TARG)
            t1 = time.perf_counter()
            dfq.bias_correction(graph, bottoms, bench.TARG)
            t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
    print('scope rep %d: LE %.2f ms | BC %.2f ms | write-back %.2f ms | total %.2f ms %s | plan cache %s' % (
        i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3, {k: round(v, 2) for k, v in spans.items()}, dfq.plan_cache_stats))
