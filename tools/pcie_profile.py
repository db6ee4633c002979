#!/usr/bin/env python3
"""Where the wall time of the drop-in calls on a CPU-resident model goes (cProfile of the third model's calls; run on the GPU box).
usage: tools/pcie_profile.py [net] [persist 0/1]"""
import cProfile, io, pstats, sys, time, os
sys.path.insert(0, '.')
if len(sys.argv) > 2:
    os.environ['DFQ_STAGE_PERSIST'] = sys.argv[2]
import torch
import torch.nn as nn
from dfq_amd import dfq, synthetic, _ffi
from dfq_amd.utils import layer_transform as lt, relation as rel
TARG = [nn.Conv2d, nn.Linear]
net = sys.argv[1] if len(sys.argv) > 1 else 'mobilenet_v2'
for rep in range(4):
    model, graph, bottoms = synthetic.build(net, seed=0)
    pr = cProfile.Profile() if rep == 3 else None
    t0 = time.perf_counter()
    lt.merge_batchnorm(model, graph, bottoms, TARG)
    t1 = time.perf_counter()
    rels = rel.create_relation(graph, bottoms, TARG, delete_single=False)
    t2 = time.perf_counter()
    if pr: pr.enable()
    dfq.cross_layer_equalization(graph, rels, TARG)
    torch.cuda.synchronize()
    if pr: pr.disable()
    t3 = time.perf_counter()
    dfq.bias_correction(graph, bottoms, TARG)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    st = getattr(_ffi._ambient, 'persist', None)
    print('rep %d: merge %.2f  le %.2f  bc %.2f ms   plan cache %s   persistent stage: %s' % (rep, (t1 - t0) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, dfq.plan_cache_stats,
          (len(st._packs), [len(p[1]) for p in st._packs], len(st._shadow), [tuple(t.shape) for t, _ in st._shadow[:6]]) if st else None), file=sys.stderr)
    if pr:
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
        print(s.getvalue()[:9000], file=sys.stderr)
