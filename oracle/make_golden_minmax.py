"""Pin the oracle's set_quant_minmax against the UNMODIFIED reference and write tests/golden/minmax_*.npz.

Runs only in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_minmax.py

For every case: build the synthetic net, fold BN with the reference's merge_batchnorm, replace every
conv / linear of the graph dict by the reference's QConv2d / QLinear (what its switch_layers does, without
the absent PyTransformer), run the reference's set_quant_minmax (utils/layer_transform.py:347-609) with a
stub for the tensor-op registry it reads through a module global, read back every layer's
quant.running_min / running_max; run the oracle on the same inputs; assert agreement; store inputs (BN
proxies, the weights case (d) needs) and the reference's ranges.
"""
from __future__ import annotations

import os
import sys
from collections import OrderedDict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)
sys.dont_write_bytecode = True

import torch                                   # noqa: E402,F401  (the reference modules expect torch to be imported)
import torch.nn as nn                          # noqa: E402

from utils import quantize as ref_q            # noqa: E402  (reference)
from utils import layer_transform as ref_lt    # noqa: E402  (reference)

from oracle import dfq_oracle as orc           # noqa: E402
from oracle import graphspec                   # noqa: E402
from dfq_amd import synthetic                  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
TARG = [nn.Conv2d, nn.Linear]
F32 = np.float32


class _NoTensorOps:
    """module_tensor_op of layer_transform.py (set by replace_op in the reference): never matches."""

    def get_graph_name(self):
        return None


def q_graph(graph):
    """graph dict with every conv / linear replaced by the reference's Q classes (same parameters)."""
    out = OrderedDict()
    for k, m in graph.items():
        if type(m) == nn.Conv2d:
            q = ref_q.QConv2d(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, m.dilation, m.groups,
                              m.bias is not None)
            q.weight.data.copy_(m.weight.data)
            if m.bias is not None:
                q.bias.data.copy_(m.bias.data)
            out[k] = q
        elif type(m) == nn.Linear:
            q = ref_q.QLinear(m.in_features, m.out_features, m.bias is not None)
            q.weight.data.copy_(m.weight.data)
            if m.bias is not None:
                q.bias.data.copy_(m.bias.data)
            out[k] = q
        else:
            out[k] = m
    return out


def tensor_op_nodes(graph, bottoms):
    """{key: number of quantisers} for the tensor ops the reference quantises (layer_transform.py:10-14):
    one quantiser per input of an add / cat, one for the input of torch.mean, F.interpolate and F.softmax."""
    out = OrderedDict()
    for k, m in graph.items():
        if isinstance(m, str) and k != 'Data':
            if 'add' in k or 'cat' in k:
                out[k] = len(bottoms[k])
            elif 'mean' in k or 'interpolate' in k or 'softmax' in k:
                out[k] = 1
    return out


def run(name, seed, keep_relu6=False, is_detection=False, N=6, tensor_ops=False):
    model, graph, bottoms = synthetic.build(name, seed=seed, keep_relu6=keep_relu6)
    ref_lt.merge_batchnorm(model, graph, bottoms, TARG)
    spec = graphspec.from_torch(graph, bottoms, TARG)
    gq = q_graph(graph)
    ops = tensor_op_nodes(graph, bottoms) if tensor_ops else OrderedDict()
    if ops:
        # the reference's own container (layer_transform.py:186-228) with the names switch_layers would give it:
        # (graph name of the op, '<op>_<line>_<number of quantisers>'), in graph order
        op_quant = [ref_q.QuantMeasure(num_bits=8, momentum=0.1) for k in ops for _ in range(ops[k])]
        names = [(k, 'op_{}_{}'.format(i, ops[k])) for i, k in enumerate(ops)]
        ref_lt.module_tensor_op = ref_lt.CustomTensorOP(op_quant, names)
    else:
        ref_lt.module_tensor_op = _NoTensorOps()
    ref_lt.set_quant_minmax(gq, bottoms, is_detection=is_detection, N=N, verbose=False)
    keys = list(graph.keys())
    want = OrderedDict()
    qi = 0
    for k, m in gq.items():
        if hasattr(m, 'quant') and bottoms[k] is not None:
            want[k] = (float(m.quant.running_min), float(m.quant.running_max))
        elif k in ops:
            want[k] = [(float(op_quant[qi + j].running_min), float(op_quant[qi + j].running_max)) for j in range(ops[k])]
            qi += ops[k]
    got = orc.set_quant_minmax(spec, is_detection=is_detection, N=N, tensor_ops=ops)
    assert list(got.keys()) == list(want.keys()), (list(got.keys()), list(want.keys()))
    worst = 0.0
    flat_want = []
    for k in want:
        pairs_g = got[k] if isinstance(want[k], list) else [got[k]]
        pairs_w = want[k] if isinstance(want[k], list) else [want[k]]
        assert len(pairs_g) == len(pairs_w)
        for pg, pw in zip(pairs_g, pairs_w):
            flat_want.append(pw)
            for a, b in zip(pg, pw):
                err = abs(a - b) / max(1.0, abs(b))
                worst = max(worst, err)
                assert err <= 1e-5, '{} {}: oracle {} reference {}'.format(name, k, got[k], want[k])
    out = {'cfg': np.array([int(keep_relu6), int(is_detection), N]),
           'ranges': np.array(flat_want, dtype=np.float64),
           'layers': np.array([keys.index(k) for k in want]),
           'counts': np.array([len(want[k]) if isinstance(want[k], list) else 0 for k in want])}   # 0: a layer, n: a tensor op
    for i, k in enumerate(spec.order):
        n = spec.nodes[k]
        if n.kind == 'bn':
            out['bn{}'.format(i)] = np.stack([n.fake_weight, n.fake_bias])
        elif n.kind == 'targ' and name in ('tiny_head', 'tiny_seg'):       # only case (d) reads weights; keep the fixtures small
            out['w{}'.format(i)] = n.weight
            if n.bias is not None:
                out['b{}'.format(i)] = n.bias
    tag = 'minmax_{}_s{}{}{}{}'.format(name, seed, '_relu6' if keep_relu6 else '', '_det' if is_detection else '',
                                       '_ops' if tensor_ops else '')
    np.savez_compressed(os.path.join(GOLD, tag + '.npz'), **out)
    print('{}: {} quantised nodes ({} tensor ops), oracle vs reference max rel err {:.2e}'.format(tag, len(want), len(ops), worst))


def kat_zeroq_rows():
    """ZeroQ's per-channel quantiser: the reference's AsymmetricQuantFunction driven the way its Quant_Conv2d /
    Quant_Linear do (per-row min / max), on weights of several shapes and bit widths."""
    import importlib.util
    spec_ = importlib.util.spec_from_file_location('zeroq_quant_utils',
                                                   os.path.join(REF, 'ZeroQ', 'utils', 'quantization_utils', 'quant_utils.py'))
    zq = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(zq)
    gen = torch.Generator().manual_seed(99)
    out = {}
    cases = [((12, 5, 3, 3), 8), ((7, 130), 8), ((16, 1, 5, 5), 4), ((9, 40), 6), ((5, 3, 1, 1), 2)]
    for i, (shape, bits) in enumerate(cases):
        w = torch.randn(*shape, generator=gen) * (torch.rand(shape[0], *([1] * (len(shape) - 1)), generator=gen) + 0.05)
        if i == 1:
            w[2] = 0.25                                   # constant row: span clamps at 1e-8
        flat = w.contiguous().view(shape[0], -1)
        ref = zq.AsymmetricQuantFunction.apply(w, bits, flat.min(dim=1).values, flat.max(dim=1).values)
        got = orc.zeroq_quant_rows(w.numpy(), bits)
        same = (np.ascontiguousarray(got).view(np.int32) == ref.numpy().view(np.int32)) | (np.isnan(got) & np.isnan(ref.numpy()))
        assert same.all(), 'zeroq case {}: {} elements differ'.format(i, int((~same).sum()))
        out['x{}'.format(i)] = w.numpy()
        out['y{}'.format(i)] = ref.numpy()
        out['bits{}'.format(i)] = np.array(bits)
    np.savez_compressed(os.path.join(GOLD, 'kat_zeroq_rows.npz'), **out)
    print('kat_zeroq_rows: {} cases, oracle bit-exact with the reference'.format(len(cases)))


def main():
    os.makedirs(GOLD, exist_ok=True)
    kat_zeroq_rows()
    run('tiny_mobile', 0)
    run('tiny_mobile', 1, keep_relu6=True)
    run('tiny_res', 0)
    run('tiny_res', 2, keep_relu6=True)
    run('tiny_cat', 0, is_detection=True)
    run('tiny_cat', 1, keep_relu6=True)
    run('tiny_wide', 3, keep_relu6=True)
    run('tiny_head', 0)
    run('tiny_head', 4, keep_relu6=True)
    run('tiny_res', 0, tensor_ops=True)
    run('tiny_res', 2, keep_relu6=True, tensor_ops=True)
    run('tiny_mobile', 0, tensor_ops=True)
    run('tiny_cat', 1, keep_relu6=True, tensor_ops=True)
    run('resnet18', 0, tensor_ops=True)
    run('tiny_seg', 0, tensor_ops=True)
    run('tiny_seg', 1, keep_relu6=True, tensor_ops=True)
    run('tiny_seg', 2)


if __name__ == '__main__':
    main()
