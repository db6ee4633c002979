"""Pin config 5 (--distill_range) against the UNMODIFIED reference and write tests/golden/range_*.npz.

Runs only in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_range.py

`improve_dfq.py` imports two packages that are absent here (the un-vendored PyTransformer submodule and
tensorboardX, improve_dfq.py:5,9); neither is used by the three functions on the path, so they are injected
into sys.modules as empty stubs and the reference module is imported as it is.  The only other
change to the environment: `update_quant_range` moves every batch with `.cuda()` (improve_dfq.py:284) and this
container has no GPU, so `torch.Tensor.cuda` is the identity while the reference runs.

For every case: build a small network out of the REFERENCE's quantised layer classes, run

    set_update_stat(model, [QuantMeasure], True)        improve_dfq.py:299-309
    update_quant_range(model, data, graph, bottoms)     improve_dfq.py:280-297
    set_update_stat(model, [QuantMeasure], False)
    y = model(data[0])                                  eval forward with the recorded ranges

and store: the parameters, the batches, the activation every QuantMeasure saw for every batch (so that the
range kernels can be checked on bit-identical inputs even where the convolution in front of them is computed
by a different library), every running_min / running_max before and after the first-layer pin, and y.  The
numpy oracle (oracle.quant_measure_forward) is asserted against the same records.
"""
from __future__ import annotations

import os
import sys
import types
from collections import OrderedDict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)
sys.dont_write_bytecode = True

import torch                                   # noqa: E402
import torch.nn as nn                          # noqa: E402

# ---- stubs for the two absent imports of improve_dfq.py (no attribute of them is touched on this path) ----
for name in ('PyTransformer', 'PyTransformer.transformers', 'PyTransformer.transformers.torchTransformer', 'tensorboardX'):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules['PyTransformer.transformers.torchTransformer'].TorchTransformer = type('TorchTransformer', (), {})
sys.modules['tensorboardX'].SummaryWriter = type('SummaryWriter', (), {})

import improve_dfq as ref_imp                  # noqa: E402  (reference, unmodified)
from utils import quantize as ref_q            # noqa: E402  (reference)

from oracle import dfq_oracle as orc           # noqa: E402
sys.path.insert(2, os.path.join(ROOT, 'tests'))
from common import build_range_net, RANGE_LAYERS as LAYERS   # noqa: E402  (the same builder the tests use)

GOLD = os.path.join(ROOT, 'tests', 'golden')
F32 = np.float32


def run_case(kind, seed, n_batches, is_detection=False):
    g = torch.Generator().manual_seed(seed)
    net, graph, bottoms = build_range_net(ref_q, kind)
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.4 if p.dim() > 1 else 0.1))
    data = [torch.randn(4, 3, 10, 10, generator=g).clamp_(-2.1179, 2.64) for _ in range(n_batches)]
    out = {'cfg': np.array([n_batches, int(is_detection)])}
    for k, v in net.state_dict().items():
        if 'running_' not in k:
            out['param.' + k] = v.numpy().copy()
    for i, b in enumerate(data):
        out['data{}'.format(i)] = b.numpy().copy()

    # record what every QuantMeasure sees, in call order
    seen = {k: [] for k in LAYERS}
    hooks = [graph[k].quant.register_forward_pre_hook(lambda m, a, k=k: seen[k].append(a[0].detach().numpy().copy()))
             for k in LAYERS]
    ref_imp.set_update_stat(net, [ref_q.QuantMeasure], True)
    assert all(graph[k].quant.update_stat for k in LAYERS)
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        ref_imp.update_quant_range(net, data, graph, bottoms, is_detection=is_detection)
    finally:
        torch.Tensor.cuda = orig_cuda
    ref_imp.set_update_stat(net, [ref_q.QuantMeasure], False)
    assert not any(graph[k].quant.update_stat for k in LAYERS)
    for h in hooks:
        h.remove()
    for k in LAYERS:
        qm = graph[k].quant
        out['range.' + k] = np.array([float(qm.running_min), float(qm.running_max)], dtype=F32)
        for i, a in enumerate(seen[k]):
            out['act.{}.{}'.format(k, i)] = a
    with torch.no_grad():
        y = net(data[0])
    out['y'] = y.numpy().copy()

    # ---- the oracle on the same records ----
    for k in LAYERS:
        rmin, rmax = F32(0.0), F32(0.0)
        for a in seen[k]:
            _, rmin, rmax = orc.quant_measure_forward(a, rmin, rmax, update_stat=True)
        if k == 'c0':
            rmin, rmax = (F32(-1.0), F32(1.0)) if is_detection else (F32(-2.11790393), F32(2.64))
        got = out['range.' + k]
        # The per-batch value is mean_n(max_chw x[n]) in float32 (quantize.py:106-107).  torch sums the n per-sample
        # extrema in float32 in an order that depends on n and on the SIMD width of the build (and is different
        # again on the GPU, where the reference runs this step); the oracle and the engine round the exact mean
        # once.  The two agree to 1 ulp (observed: identical for ~35-70 % of random inputs), far inside the 1e-5 float32 contract.
        for a, b in ((got[0], rmin), (got[1], rmax)):
            assert abs(float(a) - float(b)) <= 1e-6 * max(1.0, abs(float(a))), \
                '{} {}: oracle range ({}, {}) != reference {}'.format(kind, k, rmin, rmax, got)
        out['oracle_range.' + k] = np.array([rmin, rmax], dtype=F32)
    tag = 'range_{}_s{}{}'.format(kind, seed, '_det' if is_detection else '')
    np.savez_compressed(os.path.join(GOLD, tag + '.npz'), **out)
    print('{}: {} batches, ranges {}'.format(tag, n_batches, {k: out['range.' + k].tolist() for k in LAYERS}))


def main():
    os.makedirs(GOLD, exist_ok=True)
    run_case('plain', 0, 2)
    run_case('plain', 1, 3, is_detection=True)
    run_case('wq', 2, 2)


if __name__ == '__main__':
    main()
