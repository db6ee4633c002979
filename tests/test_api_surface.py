"""The rest of the reference's call surface on both backends: scale folding
(transform_quant_layer / merge_scale_to_weight), distilled-range calibration (set_update_stat /
update_quant_range / QuantMeasure inside the layers), the Q* layer classes, the sharded rebuild."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import dfq_oracle as orc
from dfq_amd import improve_dfq, sharded
from dfq_amd.utils import quantize as q
from dfq_amd.utils.relation import Relation

from common import F32, assert_bitexact, assert_close, npy


def test_merge_scale_to_weight_conv_and_linear(engine):
    rng = np.random.default_rng(7)
    conv = q.QConv2d(8, 12, 3, groups=2, bias=True).to(engine.device)
    lin = q.QLinear(12, 5).to(engine.device)
    w = rng.standard_normal((12, 4, 3, 3)).astype(F32)
    b = rng.standard_normal(12).astype(F32)
    sc = rng.uniform(0.5, 2, 12).astype(F32)
    sp = rng.uniform(0.5, 2, 8).astype(F32)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(w))
        conv.bias.copy_(torch.from_numpy(b))
    conv.set_scale(scale=engine.to(torch.from_numpy(sc.copy())), scale_prev=engine.to(torch.from_numpy(sp.copy()).view(-1, 1, 1, 1)))
    conv.merge_scale_to_weight()
    w_o, b_o = orc.merge_scale_to_weight(w, b, sc, sp, groups=2)
    assert_bitexact(npy(conv.weight), w_o, 'conv weight')
    assert_bitexact(npy(conv.bias), b_o, 'conv bias')

    wl = rng.standard_normal((5, 12)).astype(F32)
    bl = rng.standard_normal(5).astype(F32)
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(wl))
        lin.bias.copy_(torch.from_numpy(bl))
    sl = rng.uniform(0.5, 2, 5).astype(F32)
    spl = rng.uniform(0.5, 2, 12).astype(F32)
    lin.set_scale(scale=engine.to(torch.from_numpy(sl.copy())), scale_prev=engine.to(torch.from_numpy(spl.copy())))
    lin.merge_scale_to_weight()
    w_o, b_o = orc.merge_scale_to_weight(wl, bl, sl, spl, linear=True)
    assert_bitexact(npy(lin.weight), w_o, 'linear weight')
    assert_bitexact(npy(lin.bias), b_o, 'linear bias')


def _kat_merge_names():
    import os
    from common import GOLD
    return [str(n) for n in np.load(os.path.join(GOLD, 'kat_merge_scale.npz'))['names']]


@pytest.mark.parametrize('name', _kat_merge_names())
def test_merge_scale_to_weight_against_reference(engine, name):
    """Row a11 pinned: tests/golden/kat_merge_scale.npz holds what the REFERENCE's QConv2d / QLinear
    .set_scale + .merge_scale_to_weight (utils/quantize.py:136-174, 262-289) leave in weight and bias
    (oracle/make_golden.py:kat_merge_scale) -- grouped, depthwise, scale only, scale_prev only, no bias; QLinear's
    merge_scale_prev multiplies.  Contract: bit-exact (one IEEE divide, one multiply per element, in that order)."""
    import os
    from common import GOLD
    g = np.load(os.path.join(GOLD, 'kat_merge_scale.npz'))
    linear, groups = [int(v) for v in g[name + '.cfg']]
    w = g[name + '.w']
    has_b = (name + '.b') in g.files
    if linear:
        layer = q.QLinear(w.shape[1], w.shape[0], bias=has_b)
    else:
        layer = q.QConv2d(w.shape[1] * groups, w.shape[0], w.shape[2], groups=groups, bias=has_b)
    layer = layer.to(engine.device)
    with torch.no_grad():
        layer.weight.copy_(torch.from_numpy(w))
        if has_b:
            layer.bias.copy_(torch.from_numpy(g[name + '.b']))
    sc = engine.to(torch.from_numpy(g[name + '.scale'].copy())) if (name + '.scale') in g.files else None
    sp = None
    if (name + '.scale_prev') in g.files:
        sp = torch.from_numpy(g[name + '.scale_prev'].copy())
        sp = engine.to(sp.view(-1, 1) if linear else sp.view(-1, 1, 1, 1))
    layer.set_scale(scale=sc, scale_prev=sp)
    layer.merge_scale_to_weight()
    assert_bitexact(npy(layer.weight), g[name + '.w_out'], name + ' weight')
    if has_b:
        assert_bitexact(npy(layer.bias), g[name + '.b_out'], name + ' bias')
    assert getattr(layer, 'scale', None) is None and getattr(layer, 'scale_prev', None) is None
    # and the oracle agrees with the reference on the same inputs (the oracle is what other tests use)
    w_o, b_o = orc.merge_scale_to_weight(w, g[name + '.b'] if has_b else None,
                                         g[name + '.scale'] if sc is not None else None,
                                         g[name + '.scale_prev'] if sp is not None else None, groups=groups, linear=bool(linear))
    assert_bitexact(w_o, g[name + '.w_out'], name + ' oracle weight')


def test_transform_quant_layer_swaps_and_folds(engine):
    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = q.QConv2d(3, 6, 3, padding=1)
            self.bn = nn.BatchNorm2d(6)
            self.b = q.QConv2d(6, 4, 1)
            self.fc = q.QLinear(4, 2)

        def forward(self, x):
            x = self.b(torch.relu(self.bn(self.a(x))))
            return self.fc(x.mean((2, 3)))
    net = Net().to(engine.device).eval()
    graph = {'Data': 'Data', 'a': net.a, 'bn': net.bn, 'b': net.b, 'fc': net.fc}
    s = torch.rand(6, device=engine.device) + 0.5
    wa, wb = npy(net.a.weight).copy(), npy(net.b.weight).copy()
    net.a.set_scale(scale=s.clone())
    net.b.set_scale(scale_prev=s.clone().view(-1, 1, 1, 1))
    rels = [Relation('a', 'b', 'bn')]
    out = improve_dfq.transform_quant_layer(net, graph, rels, trainable=False)
    assert out is net
    assert type(net.a) is q.QuantNConv2d and type(net.b) is q.QuantNConv2d and type(net.fc) is q.QuantNLinear
    assert graph['a'] is net.a and graph['b'] is net.b
    assert not hasattr(net.a, 'scale') or getattr(net.a, 'scale', None) is None
    assert_bitexact(npy(net.a.weight), (wa * npy(s).reshape(-1, 1, 1, 1)).astype(F32))
    assert_bitexact(npy(net.b.weight), (wb / npy(s).reshape(1, -1, 1, 1)).astype(F32))
    net(torch.randn(2, 3, 8, 8, device=engine.device))          # still runs


def test_update_quant_range_records_activation_ranges(engine):
    net = nn.Sequential(q.QuantNConv2d(3, 4, 3, padding=1), nn.ReLU(), q.QuantNConv2d(4, 4, 1)).to(engine.device).eval()
    graph = {'Data': 'Data', 'c0': net[0], 'r': net[1], 'c1': net[2]}
    bottoms = {'Data': None, 'c0': ['Data'], 'r': ['c0'], 'c1': ['r']}
    improve_dfq.set_update_stat(net, [q.QuantMeasure], True)
    assert net[0].quant.update_stat and net[2].quant.update_stat
    rng = np.random.default_rng(0)
    data = [torch.from_numpy(rng.standard_normal((4, 3, 6, 6)).astype(F32)) for _ in range(2)]
    # expected running range of the second layer's input = running max / min over batches of the per-batch
    # mean-of-per-sample extrema (quantize.py:106-107), from the oracle on the activations the module saw
    seen = []
    hook = net[2].quant.register_forward_pre_hook(lambda m, a: seen.append(npy(a[0])))
    improve_dfq.update_quant_range(net, data, graph, bottoms)
    hook.remove()
    improve_dfq.set_update_stat(net, [q.QuantMeasure], False)
    assert float(net[0].quant.running_max) == pytest.approx(2.64) and float(net[0].quant.running_min) == pytest.approx(-2.11790393)
    rmin, rmax = F32(0.0), F32(0.0)
    assert len(seen) == 2
    for a in seen:
        _, rmin, rmax = orc.quant_measure_forward(a, rmin, rmax, update_stat=True)
    assert_bitexact(npy(net[2].quant.running_min), np.array([rmin], dtype=F32), 'running_min')
    assert_bitexact(npy(net[2].quant.running_max), np.array([rmax], dtype=F32), 'running_max')
    assert rmax > 0.0 and rmin == 0.0        # the input of the second layer comes out of a ReLU
    y = net(data[0].to(engine.device))
    assert y.shape == (4, 4, 6, 6) and torch.isfinite(y).all()


def test_qconv_forward_matches_manual_fake_quant(engine):
    torch.manual_seed(0)
    layer = q.QuantConv2d(3, 5, 3, padding=1).to(engine.device).eval()
    x = torch.randn(2, 3, 7, 7, device=engine.device)
    layer.quant.running_min.fill_(-2.0)
    layer.quant.running_max.fill_(2.0)
    y = layer(x)
    xq = torch.from_numpy(orc.uniform_quantize(npy(x), 8, -2.0, 2.0))
    w = npy(layer.weight)
    wq = torch.from_numpy(orc.uniform_quantize(w, 8, float(w.min()), float(w.max())))
    bq = torch.from_numpy(orc.uniform_quantize(npy(layer.bias), 16))
    want = torch.nn.functional.conv2d(xq, wq, bq, padding=1)
    assert_close(npy(y), want.numpy(), 'QuantConv2d forward', tol=1e-5)


def test_sharded_rebuild_kernels(engine):
    """diag(S_out) . W0 . diag(1/S_in) with the engine's batched rebuild launch (dfq_amd/sharded.py): ONE launch over
    weights of every geometry (vector and scalar paths, grouped, linear, rows that are no multiple of 4, an out-of-place
    copy item, an unaligned view) plus the per-channel vectors; two separately rounded float32 operations per element."""
    rng = np.random.default_rng(3)
    items, want = [], []
    cases = [((12, 6, 3, 3), 1), ((12, 3, 3, 3), 4), ((10, 7), 1), ((16, 8, 1, 1), 1), ((24, 1, 3, 3), 24), ((9, 4100), 1),
             ((8, 16, 3, 3), 2)]
    for shape, groups in cases:
        w = rng.standard_normal(shape).astype(F32)
        b = rng.standard_normal(shape[0]).astype(F32)
        so = rng.uniform(0.5, 2, shape[0]).astype(F32)
        si = rng.uniform(0.5, 2, shape[1] * groups).astype(F32)
        tw, tb = engine.to(torch.from_numpy(w.copy())), engine.to(torch.from_numpy(b.copy()))
        tso, tsi = engine.to(torch.from_numpy(so)), engine.to(torch.from_numpy(si))
        items += [(tw, tw, tso, tsi, groups), (tb, tb, tso, None, 1)]
        t = (w * so.reshape((-1,) + (1,) * (w.ndim - 1))).astype(F32)
        per_row = si.reshape(groups, -1).repeat(shape[0] // groups, axis=0)
        want += [(t / per_row.reshape((shape[0], shape[1]) + (1,) * (w.ndim - 2))).astype(F32), (b * so).astype(F32)]
    # rows only / columns only / plain copy into another buffer / a view that is not 16-byte aligned
    w = rng.standard_normal((6, 8)).astype(F32)
    s6, s8 = rng.uniform(0.5, 2, 6).astype(F32), rng.uniform(0.5, 2, 8).astype(F32)
    src = engine.to(torch.from_numpy(w.copy()))
    d0, d1, d2 = (torch.zeros_like(src) for _ in range(3))
    arena = engine.to(torch.zeros(49 + 3))
    d3 = arena[3:51].view(6, 8)
    items += [(src, d0, engine.to(torch.from_numpy(s6)), None, 1), (src, d1, None, engine.to(torch.from_numpy(s8)), 1),
              (src, d2, None, None, 1), (src, d3, None, None, 1)]
    want += [(w * s6[:, None]).astype(F32), (w / s8[None, :]).astype(F32), w, w]
    plan = sharded._RebuildPlan(items)
    assert plan.elements == sum(int(np.prod(x.shape)) for x in want)
    plan.run()
    for (_, dst, _, _, _), ref in zip(items, want):
        assert_bitexact(npy(dst), ref, 'rebuild item {}'.format(tuple(ref.shape)))
    assert_bitexact(npy(src), w, 'out-of-place source untouched')
    plan.close()


@pytest.mark.gpu
def test_scale_rows_and_cols_beyond_65535_output_channels():
    """A Linear with more than 65535 output channels (a large classifier / embedding): the row and column rescale
    kernels use a 1-D grid (a 2-D grid capped the row count)."""
    from dfq_amd import _ffi
    dev = torch.device('cuda', 0)
    rows, cols = 70001, 6
    g = torch.Generator().manual_seed(0)
    w = torch.randn(rows, cols, generator=g)
    so, si = torch.rand(rows, generator=g) + 0.5, torch.rand(cols, generator=g) + 0.5
    tw, tso, tsi = w.to(dev), so.to(dev), si.to(dev)
    lib = _ffi.lib()
    _ffi.check(lib.dfq_scale_rows(_ffi.ptr(tw), rows, cols, _ffi.ptr(tso), 0, _ffi.stream_arg()))
    _ffi.check(lib.dfq_scale_cols(_ffi.ptr(tw), rows, cols, 1, 1, _ffi.ptr(tsi), 1, _ffi.stream_arg()))
    want = ((w * so.view(-1, 1)) / si.view(1, -1)).numpy()
    assert_bitexact(npy(tw), want, 'weight of a 70001-row linear')


def test_plan_cache_of_the_drop_in_entry_points(engine):
    """Calling cross_layer_equalization / bias_correction again on the same (device-resident) graph reuses the plans
    (the drop-in API rebuilt ~12 device allocations per call); results are those of fresh plans."""
    from dfq_amd import dfq, synthetic
    from dfq_amd.utils import layer_transform as lt
    from dfq_amd.utils import relation as rel
    from common import TARG, snapshot
    dfq.clear_plan_cache()
    for k in dfq.plan_cache_stats:
        dfq.plan_cache_stats[k] = 0
    model, graph, bottoms = synthetic.build('tiny_mobile', seed=0)
    model.to(engine.device)
    lt.merge_batchnorm(model, graph, bottoms, TARG)
    rels = rel.create_relation(graph, bottoms, TARG)
    dfq.cross_layer_equalization(graph, rels, TARG, max_sweeps=3)
    dfq.cross_layer_equalization(graph, rels, TARG, max_sweeps=2)          # same graph again: 5 sweeps in all
    dfq.bias_correction(graph, bottoms, TARG)
    dfq.bias_correction(graph, bottoms, TARG)
    cached = (snapshot(graph), [npy(r.get_scale_vec()) for r in rels])
    if engine.kind == 'gpu':                                                # CPU tensors are never cached (shadow copies)
        assert dfq.plan_cache_stats == {'le_hits': 1, 'le_misses': 1, 'bc_hits': 1, 'bc_misses': 1}
    # reference run without any cache hit
    dfq.clear_plan_cache()
    model2, graph2, bottoms2 = synthetic.build('tiny_mobile', seed=0)
    model2.to(engine.device)
    lt.merge_batchnorm(model2, graph2, bottoms2, TARG)
    rels2 = rel.create_relation(graph2, bottoms2, TARG)
    dfq.cross_layer_equalization(graph2, rels2, TARG, max_sweeps=3)
    dfq.clear_plan_cache()
    dfq.cross_layer_equalization(graph2, rels2, TARG, max_sweeps=2)
    dfq.clear_plan_cache()
    dfq.bias_correction(graph2, bottoms2, TARG)
    dfq.clear_plan_cache()
    dfq.bias_correction(graph2, bottoms2, TARG)
    fresh = (snapshot(graph2), [npy(r.get_scale_vec()) for r in rels2])
    for k in fresh[0]:
        assert_bitexact(cached[0][k], fresh[0][k], k)
    for a, b in zip(cached[1], fresh[1]):
        assert_bitexact(a, b, 'cumulative S')
    dfq.clear_plan_cache()
