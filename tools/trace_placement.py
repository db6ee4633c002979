#!/usr/bin/env python
"""Which CU does every tile of the persistent equalisation launch run on?  (dfq_le_resident_trace: XCC_ID / HW_ID of every workgroup.)
   python tools/trace_placement.py [net]      prints, per CU, the layers of the tiles it hosts; then how many CUs host 0/1/2/3 tiles of
   the `big` layers (the last three paired layers: the ones that pace a sweep of MobileNetV2)."""
import collections
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn as nn

from dfq_amd import _ffi, dfq, synthetic
from dfq_amd.utils import layer_transform as lt
from dfq_amd.utils import relation as rel

TARG = [nn.Conv2d, nn.Linear]
net = sys.argv[1] if len(sys.argv) > 1 else 'mobilenet_v2'
dev = torch.device('cuda', 0)
model, graph, bottoms = synthetic.build(net, seed=0)
model.to(dev)
lt.merge_batchnorm(model, graph, bottoms, TARG)
rels = rel.create_relation(graph, bottoms, TARG)
plan = dfq.build_le_plan(graph, rels, TARG)
cfg = dfq._le_config((1e-8, 1e8), -1.0, 10 ** 9, False, 0, None)
n = _ffi.lib().dfq_le_resident_trace_words(plan._plan)
out = (ctypes.c_int64 * n)()
_ffi.check(_ffi.lib().dfq_le_resident_trace(plan._plan, ctypes.byref(cfg), 6, _ffi.stream_arg(), out, n))
per = 6 * 16
where = collections.defaultdict(list)
order = []
for t in range(n // per):
    meta, hw = int(out[t * per + 7]), int(out[t * per + 16 + 7])
    layer = meta >> 32
    xcc, cu, sh, se = hw >> 32, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    where[(xcc, se, sh, cu)].append(layer)
    order.append((t, layer, xcc, se, sh, cu))
layers = sorted({l for v in where.values() for l in v})
big = set(layers[-3:])
print('workgroup -> (layer, xcc, se, sh, cu), first 40:', order[:40])
hist = collections.Counter(sum(1 for l in v if l in big) for v in where.values())
print('CUs in use:', len(where), ' tiles per CU:', dict(collections.Counter(len(v) for v in where.values())))
print('CUs by number of BIG tiles (layers {}) they host:'.format(sorted(big)), dict(hist))
for k in sorted(where)[:24]:
    print(k, where[k])
