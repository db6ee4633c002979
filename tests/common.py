"""Shared helpers of the parity tests (numpy side: oracle; torch side: engine under test)."""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TARG = [nn.Conv2d, nn.Linear]
F32 = np.float32


def bits(a):
    return np.ascontiguousarray(np.asarray(a, dtype=F32)).view(np.int32)


def assert_bitexact(a, b, what=''):
    a = np.asarray(a, dtype=F32)
    b = np.asarray(b, dtype=F32)
    assert a.shape == b.shape, '{}: shape {} vs {}'.format(what, a.shape, b.shape)
    same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
    assert same.all(), '{}: {} of {} elements differ, max abs diff {}'.format(
        what, int((~same).sum()), a.size, float(np.nanmax(np.abs(a.astype(np.float64) - b))))


def assert_close(a, b, what='', tol=1e-5):
    """|a-b| <= tol * max(1, |b|): the float32 contract of BASELINE.json (1e-5)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, '{}: shape {} vs {}'.format(what, a.shape, b.shape)
    if a.size == 0:
        return 0.0
    err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    worst = float(np.nanmax(err))
    assert worst <= tol, '{}: max err {:.3e} > {:.1e}'.format(what, worst, tol)
    return worst


def npy(t):
    return t.detach().cpu().numpy().astype(F32)


def load_inputs(graph, gold, device):
    """Overwrite the random-init parameters of a synthetic net with the fixture's inputs."""
    with torch.no_grad():
        for i, k in enumerate(graph):
            m = graph[k]
            if type(m) in TARG:
                m.weight.copy_(torch.from_numpy(gold['in.L{}.w'.format(i)]))
                if 'in.L{}.b'.format(i) in gold:
                    m.bias.copy_(torch.from_numpy(gold['in.L{}.b'.format(i)]))
            elif type(m) == nn.BatchNorm2d:
                g, b, mu, var = gold['in.L{}.bn'.format(i)]
                m.weight.copy_(torch.from_numpy(g))
                m.bias.copy_(torch.from_numpy(b))
                m.running_mean.copy_(torch.from_numpy(mu))
                m.running_var.copy_(torch.from_numpy(var))


def snapshot(graph):
    snap = {}
    for i, k in enumerate(graph):
        m = graph[k]
        if type(m) in TARG:
            snap['L{}.w'.format(i)] = npy(m.weight)
            if m.bias is not None:
                snap['L{}.b'.format(i)] = npy(m.bias)
        elif type(m) == nn.BatchNorm2d and hasattr(m, 'fake_weight'):
            snap['L{}.fw'.format(i)] = npy(m.fake_weight)
            snap['L{}.fb'.format(i)] = npy(m.fake_bias)
    return snap


def load_stage(graph, gold, stage):
    """Put the reference's state after `stage` ('merge','le','abs','bc') into the modules."""
    with torch.no_grad():
        for i, k in enumerate(graph):
            m = graph[k]
            if type(m) in TARG:
                m.weight.copy_(torch.from_numpy(gold['{}.L{}.w'.format(stage, i)]).to(m.weight.device))
                kb = '{}.L{}.b'.format(stage, i)
                if kb in gold:
                    if m.bias is None:
                        m.bias = nn.Parameter(torch.zeros(m.weight.size(0), device=m.weight.device), requires_grad=False)
                    m.bias.copy_(torch.from_numpy(gold[kb]).to(m.weight.device))
                else:
                    m.bias = None
            elif type(m) == nn.BatchNorm2d and '{}.L{}.fw'.format(stage, i) in gold:
                fw = torch.from_numpy(gold['{}.L{}.fw'.format(stage, i)]).to(m.weight.device)
                fb = torch.from_numpy(gold['{}.L{}.fb'.format(stage, i)]).to(m.weight.device)
                if hasattr(m, 'fake_weight'):
                    m.fake_weight.copy_(fw)
                    m.fake_bias.copy_(fb)
                else:
                    m.register_buffer('fake_weight', fw.clone())
                    m.register_buffer('fake_bias', fb.clone())


def compare_stage(snap, gold, stage, exact=False, tol=1e-5, what=''):
    keys = [k[len(stage) + 1:] for k in gold.files if k.startswith(stage + '.')]
    assert set(keys) == set(snap), '{} {}: key sets differ: {}'.format(what, stage, set(keys) ^ set(snap))
    worst = 0.0
    for k in keys:
        if exact:
            assert_bitexact(snap[k], gold[stage + '.' + k], '{} {} {}'.format(what, stage, k))
        else:
            worst = max(worst, assert_close(snap[k], gold[stage + '.' + k], '{} {} {}'.format(what, stage, k), tol))
    return worst


NET_FIXTURES = [
    ('tiny_mobile', 0, ''), ('tiny_mobile', 1, '_abs'), ('tiny_mobile', 2, '_signed'),
    ('tiny_res', 0, ''), ('tiny_cat', 0, ''), ('tiny_cat', 3, '_abs'),
]


def net_fixture(name, seed, suffix):
    return np.load(os.path.join(GOLD, 'net_{}_s{}{}.npz'.format(name, seed, suffix)))


# ---- config 5 (--distill_range): the small network of tests/golden/range_*.npz -------------------------
RANGE_LAYERS = ('c0', 'c1', 'c2', 'fc')


def build_range_net(q, kind):
    """The same small network from either module namespace (`q` = the reference's utils.quantize, in
    oracle/make_golden_range.py, or dfq_amd.utils.quantize, in the tests).  'plain': QuantN* layers (what
    transform_quant_layer(trainable=False) leaves, main_cls.py:184); 'wq': Quant* layers, which also
    fake-quantise weights and biases in forward."""
    from collections import OrderedDict
    conv = q.QuantNConv2d if kind == 'plain' else q.QuantConv2d
    lin = q.QuantNLinear if kind == 'plain' else q.QuantLinear

    class RangeNet(nn.Module):
        def __init__(self):
            super().__init__()
            self.c0 = conv(3, 8, 3, padding=1)
            self.r0 = nn.ReLU()
            self.c1 = conv(8, 8, 3, padding=1, groups=8)
            self.r1 = nn.ReLU6()
            self.c2 = conv(8, 12, 1)
            self.fc = lin(12, 5)

        def forward(self, x):
            x = self.r0(self.c0(x))
            x = self.r1(self.c1(x))
            x = self.c2(x)
            return self.fc(x.mean(3).mean(2))
    net = RangeNet().eval()
    graph = OrderedDict([('Data', 'Data'), ('c0', net.c0), ('r0', net.r0), ('c1', net.c1), ('r1', net.r1),
                         ('c2', net.c2), ('fc', net.fc)])
    bottoms = OrderedDict([('Data', None), ('c0', ['Data']), ('r0', ['c0']), ('c1', ['r0']), ('r1', ['c1']),
                           ('c2', ['r1']), ('fc', ['c2'])])
    return net, graph, bottoms


# ---- row f2 (ZeroQ distillation): the small teacher network of tests/golden/zeroq_*.npz ----------------
def build_distill_net(gen, with_pixel_bn):
    """conv-BN-ReLU x2 (+ global pool, 1x1 conv, BN on 1x1 feature maps: the H*W == 1 branch of
    ZeroQ/distill_data.py:181-182), parameters and BN statistics drawn from `gen`."""
    layers = [nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(),
              nn.Conv2d(8, 8, 3, padding=1, stride=2), nn.BatchNorm2d(8), nn.ReLU()]
    if with_pixel_bn:
        layers += [nn.AdaptiveAvgPool2d(1), nn.Conv2d(8, 6, 1), nn.BatchNorm2d(6)]
    net = nn.Sequential(*layers).eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.copy_(torch.randn(m.weight.shape, generator=gen) * 0.3)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.1)
            elif isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.5)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)
    return net
