#!/usr/bin/env python3
"""Wall time of the pieces of the drop-in calls on a CPU-resident model (run on the GPU box): the functions are wrapped with timers.
usage: tools/pcie_breakdown.py [net]"""
import sys, time, collections
sys.path.insert(0, '.')
import torch, torch.nn as nn
from dfq_amd import dfq, synthetic, _ffi
from dfq_amd.utils import layer_transform as lt, relation as rel
TARG = [nn.Conv2d, nn.Linear]
acc = collections.OrderedDict()
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + (time.perf_counter() - t) * 1e3
    setattr(obj, name, g)
wrap(dfq, '_le_cache_key', 'le key'); wrap(dfq, '_bc_cache_key', 'bc key') if hasattr(dfq, '_bc_cache_key') else None
wrap(dfq.LEPlan, 'run', 'le run'); wrap(dfq.BCPlan, 'run', 'bc run')
wrap(_ffi.Stage, '_writeback', 'writeback'); wrap(_ffi.Stage, 'prefetch', 'prefetch'); wrap(_ffi.Stage, 'out_like_many', 'S out')
wrap(_ffi.Stage, 'begin_call', 'begin_call'); wrap(_ffi, '_to_host', 'd2h'); wrap(_ffi, '_pinned', 'pinned alloc'); wrap(_ffi, '_to_device', 'h2d'); wrap(_ffi.Stage, 'reset', 'reset'); wrap(_ffi.Stage, 'adopt', 'adopt'); wrap(torch, 'cat', 'torch.cat')
wrap(dfq, '_bc_tables_cached', 'bc tables') if hasattr(dfq, '_bc_tables_cached') else None
net = sys.argv[1] if len(sys.argv) > 1 else 'mobilenet_v2'
for rep in range(4):
    model, graph, bottoms = synthetic.build(net, seed=0)
    acc.clear(); t0 = time.perf_counter()
    lt.merge_batchnorm(model, graph, bottoms, TARG)
    t1 = time.perf_counter(); m = dict(acc); acc.clear()
    rels = rel.create_relation(graph, bottoms, TARG, delete_single=False)
    t2 = time.perf_counter()
    dfq.cross_layer_equalization(graph, rels, TARG); torch.cuda.synchronize()
    t3 = time.perf_counter(); l = dict(acc); acc.clear()
    dfq.bias_correction(graph, bottoms, TARG); torch.cuda.synchronize()
    t4 = time.perf_counter(); b = dict(acc)
    f = lambda d: ' '.join('%s %.2f' % kv for kv in d.items())
    print('rep %d\n  merge %.2f: %s\n  le %.2f: %s\n  bc %.2f: %s' % (rep, (t1 - t0) * 1e3, f(m), (t3 - t2) * 1e3, f(l), (t4 - t3) * 1e3, f(b)), file=sys.stderr)
