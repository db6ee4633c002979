"""Parity of the HIP engine with the CPU oracle and with the reference's golden outputs.

Every test runs on both backends of conftest.engine: 'emu' (kernel sources under the CPU emulation,
`-m "not gpu"`) and 'gpu' (libdfq_hip.so on a real MI355X, `-m gpu` -- the parity tests proper).

Contract (BASELINE.json north_star / DESIGN.md):
  * fake-quant round trip: bit-exact float32 outputs and integer codes;
  * LE / BC float32 weights, biases, BN proxies, S: |err| <= 1e-5 against the reference; against the
    oracle (same IEEE single operations in the same order) LE is bit-exact.
"""
import copy
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import dfq_oracle as orc
from oracle import graphspec
from dfq_amd import dfq, synthetic
from dfq_amd.utils import layer_transform as lt
from dfq_amd.utils import quantize as q
from dfq_amd.utils import relation as rel

from common import (GOLD, NET_FIXTURES, TARG, F32, assert_bitexact, assert_close, compare_stage, load_inputs,
                    load_stage, net_fixture, npy, snapshot)


# ---------------------------------------------------------------------------------------------
# fake-quant (a4)
# ---------------------------------------------------------------------------------------------
class Engine_cpu:
    """stand-in for the `engine` fixture's object when the caller's tensors must stay on the host"""
    kind = 'host'
    device = torch.device('cpu')

    def to(self, t):
        return t


def test_fake_quant_known_answers(engine):
    g = np.load(os.path.join(GOLD, 'kat_fake_quant.npz'))
    for i, (nbits, sym, mn, mx) in enumerate(g['cases']):
        x = engine.to(torch.from_numpy(g['x{}'.format(i)].copy()))
        if np.isnan(mn):
            y = q.quantize(x, num_bits=int(nbits))
        else:
            y = q.UniformQuantize.apply(x, int(nbits), float(mn), float(mx), False, bool(sym))
        assert_bitexact(npy(y), g['y{}'.format(i)], 'fake-quant case {}'.format(i))


@pytest.mark.parametrize('n,nbits,sym', [(1, 8, False), (63, 8, True), (4097, 8, False), (70001, 4, True),
                                         (12345, 16, False)])
def test_fake_quant_codes_vs_oracle(engine, n, nbits, sym):
    rng = np.random.default_rng(n)
    x = (rng.standard_normal(n) * 3).astype(F32)
    mn, mx = float(x.min()), float(x.max())
    y_o, c_o = orc.uniform_quantize(x, nbits, mn, mx, sym, return_codes=True)
    y, c = q.uniform_quantize(engine.to(torch.from_numpy(x.copy())), nbits, mn, mx, False, sym, return_codes=True)
    assert_bitexact(npy(y), y_o, 'values')
    assert np.array_equal(c.cpu().numpy(), c_o.astype(np.int32)), 'integer codes differ'


def test_fake_quant_unaligned_views(engine):
    """Pointers that are not 16-byte aligned (a view starting at element 1 / 3) take the scalar path; results must not
    depend on the path."""
    rng = np.random.default_rng(21)
    x = rng.standard_normal(5003).astype(F32)
    xd = engine.to(torch.from_numpy(x.copy()))
    for off in (1, 3):
        got = npy(q.quantize(xd[off:], 8, -2.5, 3.0))
        assert_bitexact(got, orc.uniform_quantize(x[off:], 8, -2.5, 3.0), 'offset {}'.format(off))
        mm = npy(q.tensor_minmax(xd[off:]))
        assert mm[0] == x[off:].min() and mm[1] == x[off:].max()


def test_fake_quant_inplace_and_ste(engine):
    x = engine.to(torch.linspace(-2, 2, 1000))
    ref = orc.uniform_quantize(npy(x), 8, -2.0, 2.0)
    xin = x.clone()
    out = q.quantize(xin, 8, -2.0, 2.0, inplace=True)
    assert out.data_ptr() == xin.data_ptr()
    assert_bitexact(npy(xin), ref)
    xg = x.clone().requires_grad_(True)
    q.quantize(xg, 8, -2.0, 2.0).sum().backward()
    assert torch.equal(xg.grad, torch.ones_like(xg))          # straight-through estimator


def test_tensor_minmax_and_sample_stats(engine):
    rng = np.random.default_rng(5)
    for shape in [(1,), (7, 3), (16, 5, 9, 9), (3, 100003)]:
        x = rng.standard_normal(shape).astype(F32)
        xd = engine.to(torch.from_numpy(x.copy()))
        mm = npy(q.tensor_minmax(xd))
        assert mm[0] == x.min() and mm[1] == x.max()
        n = shape[0]
        got = npy(q.sample_minmax_mean(xd, n))
        want = orc.sample_minmax_mean(x)
        assert_bitexact(got, np.array(want, dtype=F32), 'sample stats {}'.format(shape))


def test_quant_measure(engine):
    rng = np.random.default_rng(11)
    x = rng.standard_normal((8, 4, 6, 6)).astype(F32)
    m = q.QuantMeasure(update_stat=True).to(engine.device).eval()
    y = m(engine.to(torch.from_numpy(x.copy())))
    y_o, rmin, rmax = orc.quant_measure_forward(x, 0.0, 0.0, update_stat=True)
    assert_bitexact(npy(m.running_min), np.array([rmin], dtype=F32))
    assert_bitexact(npy(m.running_max), np.array([rmax], dtype=F32))
    assert_bitexact(npy(y), y_o, 'QuantMeasure output')
    m.set_update_stat(False)
    y2 = m(engine.to(torch.from_numpy((x * 0.5).astype(F32))))
    y2_o, _, _ = orc.quant_measure_forward((x * 0.5).astype(F32), rmin, rmax, update_stat=False)
    assert_bitexact(npy(y2), y2_o)


def test_quant_measure_one_launch_equals_two_launches(engine, monkeypatch):
    """Round 4: QuantMeasure.forward with range tracking is ONE launch (dfq_quant_measure_fused: persistent workgroups take the
    per-sample extrema, meet on a monotonic arrival counter, then quantise).  Against the two-launch form (DFQ_QM_FUSED=0) on a
    sequence of calls -- the counter and the slot parity carry over from call to call, shapes of one sample, of spans that do
    not divide, of a tail that is no multiple of four -- outputs and running ranges must be bit-identical, and equal the
    oracle's."""
    from dfq_amd import _ffi
    rng = np.random.default_rng(21)
    shapes = [(8, 4, 6, 6), (8, 4, 6, 6), (1, 3, 5, 7), (3, 5, 33, 31), (5, 2, 70, 71), (8, 4, 6, 6)]
    xs = [rng.standard_normal(sh).astype(F32) * (1.0 + i) for i, sh in enumerate(shapes)]
    outs = {}
    for fused in (True, False):
        monkeypatch.setattr(q, '_QM_FUSED', fused)
        m = q.QuantMeasure(update_stat=True).to(engine.device).eval()
        res = []
        for x in xs:
            y = m(engine.to(torch.from_numpy(x.copy())))
            res.append((npy(y), npy(m.running_min).copy(), npy(m.running_max).copy()))
        if fused:
            n = xs[-1].shape[0]
            _ffi.check(_ffi.lib().dfq_quant_measure_fused_status(_ffi.ptr(m._qm_scratch), n, _ffi.stream_arg()))
            assert m._qm_arrivals > 0
        outs[fused] = res
    rmin, rmax = 0.0, 0.0
    for i, (a, b) in enumerate(zip(outs[True], outs[False])):
        for u, v, what in zip(a, b, ('output', 'running_min', 'running_max')):
            assert_bitexact(u, v, 'call {}: {}'.format(i, what))
        y_o, rmin, rmax = orc.quant_measure_forward(xs[i], rmin, rmax, update_stat=True)
        assert_bitexact(a[0], y_o, 'call {}: oracle output'.format(i))
        assert_bitexact(a[1], np.array([rmin], dtype=F32), 'call {}: oracle running_min'.format(i))
        assert_bitexact(a[2], np.array([rmax], dtype=F32), 'call {}: oracle running_max'.format(i))


@pytest.mark.parametrize('update_stat', [True, False])
def test_quant_measure_straight_through_gradient(engine, update_stat):
    """UniformQuantize.backward passes the gradient through (quantize.py:79-83); QuantMeasure in eval mode must do so on
    BOTH of its paths -- the fused range-tracking launch (update_stat) and the plain quantiser (the fused
    path returned from inside no_grad and dropped the gradient).  A CPU input to a device-resident module comes back on
    the CPU."""
    rng = np.random.default_rng(12)
    x = rng.standard_normal((4, 3, 5, 5)).astype(F32)
    m = q.QuantMeasure(update_stat=True).to(engine.device).eval()
    m(engine.to(torch.from_numpy(x.copy())))                        # fix a range first
    m.set_update_stat(update_stat)
    xin = engine.to(torch.from_numpy(x.copy())).requires_grad_(True)
    y = m(xin)
    assert y.requires_grad and y.grad_fn is not None
    wgt = engine.to(torch.from_numpy(rng.standard_normal(x.shape).astype(F32)))
    (y * wgt).sum().backward()
    assert_bitexact(npy(xin.grad), npy(wgt), 'straight-through gradient')
    # forward value is still the quantised one
    y_o, _, _ = orc.quant_measure_forward(x, float(npy(m.running_min)[0]), float(npy(m.running_max)[0]), update_stat=False)
    if not update_stat:
        assert_close(npy(y), y_o, 'forward value', tol=1e-6)
    cpu_in = torch.from_numpy(x.copy())
    assert m(cpu_in).device == cpu_in.device


@pytest.mark.gpu
def test_quant_measure_at_config5_size():
    """BASELINE.json config 5 (--distill_range): one of MobileNetV2's largest activation tensors at batch 64
    ([64, 96, 112, 112] = 7.7e7 floats) through QuantMeasure with update_stat, against the numpy oracle on a
    slice-wise evaluation: per-sample max/min are exact, their float32 mean and the five-operation fake-quant
    are the same IEEE operations -> bit-exact; then the size-independent properties (idempotence of the
    quantiser on its own output, <= 256 distinct levels, range respected)."""
    dev = torch.device('cuda', 0)
    g = torch.Generator(device='cpu').manual_seed(1)
    x = torch.randn(64, 96, 112, 112, generator=g).clamp_(-2.1179, 2.64)
    m = q.QuantMeasure(update_stat=True).to(dev).eval()
    y = m(x.to(dev))
    xn = x.numpy()
    y_o, rmin, rmax = orc.quant_measure_forward(xn, 0.0, 0.0, update_stat=True)
    assert_bitexact(npy(m.running_min), np.array([rmin], dtype=F32))
    assert_bitexact(npy(m.running_max), np.array([rmax], dtype=F32))
    yn = npy(y)
    assert_bitexact(yn, y_o, 'QuantMeasure output at full size')
    m.set_update_stat(False)
    y2 = m(y)                                                    # same range, already on the grid
    assert torch.equal(y2, y)
    assert len(np.unique(yn)) <= 256 and yn.min() >= rmin - 1e-6 and yn.max() <= rmax + 1e-6


@pytest.mark.parametrize('reduction', ['sum', 'mean', 'channel', 'spatial', None])
def test_quantize_error(engine, reduction):
    rng = np.random.default_rng(3)
    w = rng.standard_normal((12, 5, 3, 3)).astype(F32)
    got = npy(dfq._quantize_error(engine.to(torch.from_numpy(w.copy())), 8, reduction))
    want = orc.quantize_error(w, 8, reduction)
    if reduction is None:
        assert_bitexact(got, want)
    else:
        assert_close(got, np.asarray(want, dtype=F32), 'reduction {}'.format(reduction), tol=1e-6)


def _kat_qe_names():
    return [str(n) for n in np.load(os.path.join(GOLD, 'kat_quant_error.npz'))['names']]


@pytest.mark.parametrize('name', _kat_qe_names())
def test_quantize_error_against_reference(engine, name):
    """Row a3 pinned for EVERY reduction: tests/golden/kat_quant_error.npz = the reference's dfq._quantize_error
    (dfq.py:8-25; oracle/make_golden.py:kat_quant_error) on conv / pointwise / depthwise / linear tensors, unsigned and
    signed.  Elementwise result bit-exact; the reduced scalars are float32 torch sums of unspecified order in the
    reference (float64 accumulation here): 1e-5 relative."""
    g = np.load(os.path.join(GOLD, 'kat_quant_error.npz'))
    w = g[name + '.w']
    signed = name.endswith('_s')
    for red in ('sum', 'mean', 'channel', 'spatial', 'none'):
        want = g['{}.{}'.format(name, red)]
        got = npy(dfq._quantize_error(engine.to(torch.from_numpy(w.copy())), 8, red, signed))
        mine = orc.quantize_error(w, 8, None if red == 'none' else red, signed)
        if red == 'none':
            assert_bitexact(got, want, '{} elementwise'.format(name))
            assert_bitexact(mine, want, '{} oracle elementwise'.format(name))
        else:
            assert got.shape == ()
            scale = max(1.0, abs(float(want)))
            assert abs(float(got) - float(want)) <= 1e-5 * scale, (name, red, float(got), float(want))
            assert abs(float(mine) - float(want)) <= 1e-5 * scale, (name, red, float(mine), float(want))


# ---------------------------------------------------------------------------------------------
# _layer_equalization (a1): every pairing geometry, dead channels, eps, signed
# ---------------------------------------------------------------------------------------------
def _kat_names():
    return [str(n) for n in np.load(os.path.join(GOLD, 'kat_le_pairs.npz'))['names']]


# the equalisation engines a single network can run on: the resident whole-loop launch; the streaming launch of one
# workgroup per tile (what batched plans use); the opt-in streaming launch of persistent workgroups (DFQ_LE_PERSIST=1;
# with 3 workgroups every workgroup walks several tiles and tiles wait for tiles of other workgroups)
# 'streaming': the default streaming engine -- free-running segments on lean tiles, four sweeps per pass (dfq_le_cf.hpp), the other
# layers on the general tiles; '-cf2' / '-cf8': other group depths; 'streaming-general': every layer on the general tiles
# (DFQ_LE_CF=0); the persistent-workgroup variants of the general tiles run without free-running segments as well
# 'streaming-fused': layers scaled along both axes are read once per sweep -- their row tiles merge the rows' statistics over the slabs
# of a row block inside the launch (DFQ_LE_FUSE=1: the default of batched plans)
LE_ENGINES = ['resident', 'resident-cf', 'streaming', 'streaming-cf2', 'streaming-cf8', 'streaming-bg2', 'streaming-bg4', 'streaming-bg8', 'streaming-general', 'streaming-fused', 'streaming-persistent', 'streaming-persistent-3wg']


def _select_le_engine(monkeypatch, le_engine):
    for k in ('DFQ_LE_RESIDENT', 'DFQ_LE_PERSIST', 'DFQ_LE_SWEEP_WGS', 'DFQ_LE_TILE_ELEMS', 'DFQ_LE_CF', 'DFQ_LE_CF_GROUP', 'DFQ_LE_CF_BG', 'DFQ_LE_FUSE', 'DFQ_RES_CF'):
        monkeypatch.delenv(k, raising=False)
    if le_engine.startswith('streaming-bg'):     # the lean launches in the background: two groups of look-ahead, a second stream (dfq_le_cf.hpp)
        monkeypatch.setenv('DFQ_LE_CF_GROUP', le_engine[len('streaming-bg'):])
        monkeypatch.setenv('DFQ_LE_CF_BG', '1')
    if le_engine == 'resident-cf':              # closed-form column statistics of the chain ends (opt-in, dfq_le_resident.hip)
        monkeypatch.setenv('DFQ_RES_CF', '1')
    if le_engine == 'streaming-fused':
        monkeypatch.setenv('DFQ_LE_FUSE', '1')
    if not le_engine.startswith('resident'):
        monkeypatch.setenv('DFQ_LE_RESIDENT', '0')
    if le_engine.startswith('streaming-cf'):
        monkeypatch.setenv('DFQ_LE_CF_GROUP', le_engine[len('streaming-cf'):])
    if le_engine in ('streaming', 'streaming-fused'):
        monkeypatch.setenv('DFQ_LE_CF_GROUP', '4')    # (the default of a LARGE single network; a small one keeps the general tiles)
    if le_engine == 'streaming-general' or le_engine.startswith('streaming-persistent'):
        monkeypatch.setenv('DFQ_LE_CF', '0')
    if le_engine.startswith('streaming-persistent'):
        monkeypatch.setenv('DFQ_LE_PERSIST', '1')
        monkeypatch.setenv('DFQ_LE_TILE_ELEMS', '1024')     # enough tiles for workgroups to walk several (the result does not depend on it)
    if le_engine == 'streaming-persistent-3wg':
        monkeypatch.setenv('DFQ_LE_SWEEP_WGS', '3')


@pytest.mark.parametrize('name', _kat_names())
@pytest.mark.parametrize('le_engine', LE_ENGINES)
def test_layer_equalization_pairs(engine, monkeypatch, name, le_engine):
    _select_le_engine(monkeypatch, le_engine)
    g = np.load(os.path.join(GOLD, 'kat_le_pairs.npz'))
    signed, eps, use_bn = g['{}.cfg'.format(name)]
    use_bn = bool(use_bn)
    arrs = {w: g['{}.in.{}'.format(name, w)].copy() for w in ('w1', 'w2', 'b1', 'bnw', 'bnb')}
    t = {k: engine.to(torch.from_numpy(v.copy())) for k, v in arrs.items()}
    W1, W2, B1, S = dfq._layer_equalization(t['w1'], t['w2'], t['b1'] if use_bn else None,
                                            t['bnw'] if use_bn else None, t['bnb'] if use_bn else None,
                                            signed=bool(signed), eps=eps)
    assert W1 is t['w1'] and W2 is t['w2']                     # in place, like the reference
    # oracle on the same inputs: identical IEEE operations -> bit-exact
    o = {k: v.copy() for k, v in arrs.items()}
    S_o = orc.layer_equalization(o['w1'], o['w2'], o['b1'] if use_bn else None, o['bnw'] if use_bn else None,
                                 o['bnb'] if use_bn else None, signed=bool(signed), eps=eps)
    assert_bitexact(npy(S), S_o, name + ' S vs oracle')
    for k in ('w1', 'w2') + (('b1', 'bnw', 'bnb') if use_bn else ()):
        assert_bitexact(npy(t[k]), o[k], '{} {} vs oracle'.format(name, k))
        assert_close(npy(t[k]), g['{}.out.{}'.format(name, k)], '{} {} vs reference'.format(name, k))
    assert_close(npy(S), g['{}.out.S'.format(name)], name + ' S vs reference')


_SHAPES = [
    ((96, 16, 1, 1), (96, 1, 3, 3)),        # expand pw -> depthwise 3x3 (thread-per-row col side)
    ((96, 1, 3, 3), (24, 96, 1, 1)),        # depthwise -> project pw (thread-per-row row side)
    ((40, 1, 5, 5), (40, 40, 1, 1)),        # depthwise 5x5 (25 floats per row)
    ((64, 24, 1, 1), (128, 1, 3, 3)),       # depthwise with channel multiplier 2
    ((300, 160, 1, 1), (77, 300, 1, 1)),    # float4 tiles, several slabs and row blocks
    ((33, 7, 3, 3), (20, 33, 3, 3)),        # odd sizes: scalar tiles
    ((48, 6, 3, 3), (96, 12, 3, 3)),        # grouped second layer (groups = 4)
    ((130, 516), (9, 130)),                 # linear -> linear
    ((128, 12, 1, 1), (200, 128, 1, 1)),    # 128 paired channels, pointwise second layer: the bootstrap's wide-column path
    ((320, 16, 1, 1), (37, 320, 1, 1)),     # ... with a partial last channel block (320 = 5 x 64)
    ((144, 24, 1, 1), (24, 144, 1, 1)),     # ... 144 = 64 + 64 + 16 paired channels (not a multiple of the block)
    ((24, 8, 1, 1), (144, 24, 1, 1)),       # ... fewer channels than one block
    ((96, 12, 1, 1), (64, 48, 1, 1)),       # grouped pointwise second layer (groups = 2): blocks straddle a group -> generic path
]


# (one pair, one sweep: a group depth or the background mode makes no difference to the tiles that run -- one engine per tile body)
_SHAPE_ENGINES = [e for e in LE_ENGINES if e not in ('resident-cf', 'streaming-cf2', 'streaming-bg2', 'streaming-bg4', 'streaming-bg8', 'streaming-persistent-3wg')]


@pytest.mark.parametrize('s1,s2', _SHAPES)
@pytest.mark.parametrize('signed', [False, True])
@pytest.mark.parametrize('le_engine', _SHAPE_ENGINES)
def test_layer_equalization_shapes(engine, monkeypatch, s1, s2, signed, le_engine):
    """One sweep of one pair over the tile kinds of both equalisation engines (a single pair would always take the
    resident launch: DFQ_LE_RESIDENT=0 forces the streaming kernel), bit-exact against the oracle."""
    _select_le_engine(monkeypatch, le_engine)
    rng = np.random.default_rng(abs(hash((s1, s2))) % (2 ** 31))
    w1 = rng.standard_normal(s1).astype(F32)
    w2 = (rng.standard_normal(s2) * 0.2).astype(F32)
    b1 = rng.standard_normal(s1[0]).astype(F32)
    bw = np.abs(rng.standard_normal(s1[0])).astype(F32)
    bb = rng.standard_normal(s1[0]).astype(F32)
    t = [engine.to(torch.from_numpy(a.copy())) for a in (w1, w2, b1, bw, bb)]
    _, _, _, S = dfq._layer_equalization(t[0], t[1], t[2], t[3], t[4], signed=signed)
    S_o = orc.layer_equalization(w1, w2, b1, bw, bb, signed=signed)
    assert_bitexact(npy(S), S_o, 'S')
    for got, want, what in zip(t, (w1, w2, b1, bw, bb), ('w1', 'w2', 'b1', 'bn_weight', 'bn_bias')):
        assert_bitexact(npy(got), want, what)


def _random_pair(rng):
    """A random valid pairing (dfq.py:29-35): first layer [O1, I1/g1, k1, k1] or linear, second layer conv (possibly grouped,
    depthwise, with channel multiplier) or linear."""
    kind = rng.integers(0, 5)
    k1 = int(rng.choice([1, 1, 3]))
    i1 = int(rng.integers(1, 40))
    if kind == 0:        # dense -> dense
        o1 = int(rng.integers(1, 300))
        o2 = int(rng.integers(1, 120))
        k2 = int(rng.choice([1, 1, 3]))
        return (o1, i1, k1, k1), (o2, o1, k2, k2)
    if kind == 1:        # -> depthwise with channel multiplier m
        o1 = int(rng.integers(1, 300))
        m = int(rng.choice([1, 1, 2]))
        k2 = int(rng.choice([3, 5]))
        return (o1, i1, k1, k1), (o1 * m, 1, k2, k2)
    if kind == 2:        # depthwise first layer -> pointwise
        o1 = int(rng.integers(1, 300))
        k = int(rng.choice([3, 5]))
        return (o1, 1, k, k), (int(rng.integers(1, 100)), o1, 1, 1)
    if kind == 3:        # grouped second layer
        g = int(rng.choice([2, 3, 4]))
        per = int(rng.integers(1, 50))
        o1 = g * per
        return (o1, i1, k1, k1), (g * int(rng.integers(1, 20)), per, 1, 1)
    o1 = int(rng.integers(1, 400))      # linear -> linear, rows of any length
    return (o1, int(rng.integers(1, 1500))), (int(rng.integers(1, 60)), o1)


@pytest.mark.parametrize('le_engine,boot_work', [('resident', None), ('streaming', None), ('streaming', 50)])
def test_layer_equalization_random_geometries(engine, monkeypatch, le_engine, boot_work):
    """Random pairings (60 per engine on the GPU; 5, or 1 for the resident launch, on the CPU emulation, which is slow) -- odd sizes, rows shorter than a vector and longer than a wave of vectors, fewer
    channels than a bootstrap block and several blocks, slices of a block shared by several workgroups -- bit-exact against
    the oracle, three sweeps each (the second and third use the statistics the first one forwarded)."""
    _select_le_engine(monkeypatch, le_engine)
    if boot_work:
        monkeypatch.setenv('DFQ_LE_BOOT_WORK', str(boot_work))
    rng = np.random.default_rng(20260926)
    for case in range(60 if engine.device.type == 'cuda' else (1 if le_engine == 'resident' else 5)):
        s1, s2 = _random_pair(rng)
        signed = bool(rng.integers(0, 2))
        w1 = rng.standard_normal(s1).astype(F32)
        w2 = (rng.standard_normal(s2) * 0.3).astype(F32)
        b1 = rng.standard_normal(s1[0]).astype(F32)
        t = [engine.to(torch.from_numpy(a.copy())) for a in (w1, w2, b1)]
        for sweep in range(3 if engine.device.type == 'cuda' else 2):
            _, _, _, S = dfq._layer_equalization(t[0], t[1], t[2], signed=signed)
            S_o = orc.layer_equalization(w1, w2, b1, signed=signed)
            what = 'case {} {} -> {} signed={} sweep {}'.format(case, s1, s2, signed, sweep)
            assert_bitexact(npy(S), S_o, what + ' S')
            assert_bitexact(npy(t[0]), w1, what + ' w1')
            assert_bitexact(npy(t[1]), w2, what + ' w2')
            assert_bitexact(npy(t[2]), b1, what + ' b1')


@pytest.mark.parametrize('le_engine', LE_ENGINES)
def test_layer_equalization_large_rows(engine, monkeypatch, le_engine):
    """Rows longer than a workgroup, tiles of one channel, 3x3 second layer (ResNet-like)."""
    _select_le_engine(monkeypatch, le_engine)
    rng = np.random.default_rng(21)
    w1 = rng.standard_normal((70, 33, 3, 3)).astype(F32)
    w2 = (rng.standard_normal((90, 70, 3, 3)) * 0.1).astype(F32)
    b1 = rng.standard_normal(70).astype(F32)
    t = [engine.to(torch.from_numpy(a.copy())) for a in (w1, w2, b1)]
    _, _, _, S = dfq._layer_equalization(t[0], t[1], t[2])
    S_o = orc.layer_equalization(w1, w2, b1)
    assert_bitexact(npy(S), S_o)
    assert_bitexact(npy(t[0]), w1)
    assert_bitexact(npy(t[1]), w2)
    assert_bitexact(npy(t[2]), b1)


# ---------------------------------------------------------------------------------------------
# whole pipeline on the tiny nets, stage by stage against the reference's outputs
# ---------------------------------------------------------------------------------------------
def _build(name, seed, gold, engine):
    model, graph, bottoms = synthetic.build(name, seed=seed)
    load_inputs(graph, gold, 'cpu')
    model.to(engine.device)
    return model, graph, bottoms


@pytest.mark.parametrize('name,seed,suffix', NET_FIXTURES)
def test_pipeline_stages(engine, name, seed, suffix):
    gold = net_fixture(name, seed, suffix)
    absorption, signed = [bool(v) for v in gold['cfg']]
    model, graph, bottoms = _build(name, seed, gold, engine)
    spec = graphspec.from_torch(graph, bottoms, TARG)          # oracle twin of the inputs
    keys = list(graph.keys())

    lt.merge_batchnorm(model, graph, bottoms, TARG)
    compare_stage(snapshot(graph), gold, 'merge', tol=1e-6, what=name)
    orc.merge_batchnorm(spec)

    rels = rel.create_relation(graph, bottoms, TARG, delete_single=False)
    assert [[keys.index(k) for k in r.get_idxs()] for r in rels] == gold['relations'].tolist()

    dfq.cross_layer_equalization(graph, rels, TARG, converge_thres=2e-7, signed=signed)
    res = dfq.last_equalization
    orels = orc.create_relation(spec)
    n_o, S_o = orc.cross_layer_equalization(spec, orels, signed=signed)
    assert res['sweeps'] == n_o == int(gold['oracle_sweeps'])
    # engine == oracle bit for bit (same IEEE operations, same order)
    osnap = _spec_snapshot(spec)
    esnap = snapshot(graph)
    for k in osnap:
        assert_bitexact(esnap[k], osnap[k], '{} LE {} vs oracle'.format(name, k))
    for r, s in zip(rels, S_o):
        assert_bitexact(npy(r.get_scale_vec()), s, 'S_cum vs oracle')
    if res['sweeps'] == int(gold['n_sweeps']):
        compare_stage(esnap, gold, 'le', what=name)
        for i, r in enumerate(rels):
            assert_close(npy(r.get_scale_vec()), gold['S{}'.format(i)], 'S{} vs reference'.format(i))

    # the later stages start from the REFERENCE's state so every stage is compared on equal inputs
    load_stage(graph, gold, 'le')
    if absorption:
        dfq.bias_absorption(graph, rels, bottoms, 3)
    compare_stage(snapshot(graph), gold, 'abs', what=name)

    load_stage(graph, gold, 'abs')
    dfq.bias_correction(graph, bottoms, TARG, signed=signed)
    compare_stage(snapshot(graph), gold, 'bc', what=name)

    load_stage(graph, gold, 'bc')
    lt.quantize_targ_layer(graph, 8, 16, TARG)
    compare_stage(snapshot(graph), gold, 'q', exact=True, what=name)


@pytest.mark.parametrize('tile_elems,boot_work', [(48, None), (500, None), (500, 40), (None, 7)])
@pytest.mark.parametrize('name,seed,suffix', [('tiny_mobile', 2, '_signed'), ('tiny_cat', 0, ''), ('tiny_res', 0, '')])
def test_equalization_tile_shapes(engine, monkeypatch, name, seed, suffix, tile_elems, boot_work):
    """Small tiles force many row slabs / row blocks / col tiles per relation, a small bootstrap work unit makes several
    workgroups share a block of channels (their statistics merge through atomicMax): the result may not depend on the
    decomposition (min/max and the scale solve are exact).  Streaming engine (the resident launch has its own tiling)."""
    monkeypatch.setenv('DFQ_LE_RESIDENT', '0')
    if tile_elems:
        monkeypatch.setenv('DFQ_LE_TILE_ELEMS', str(tile_elems))
    if boot_work:
        monkeypatch.setenv('DFQ_LE_BOOT_WORK', str(boot_work))
    gold = net_fixture(name, seed, suffix)
    signed = bool(gold['cfg'][1])
    model, graph, bottoms = _build(name, seed, gold, engine)
    spec = graphspec.from_torch(graph, bottoms, TARG)
    lt.merge_batchnorm(model, graph, bottoms, TARG)
    orc.merge_batchnorm(spec)
    rels = rel.create_relation(graph, bottoms, TARG)
    dfq.cross_layer_equalization(graph, rels, TARG, signed=signed)
    n_o, S_o = orc.cross_layer_equalization(spec, orc.create_relation(spec), signed=signed)
    assert dfq.last_equalization['sweeps'] == n_o
    osnap, esnap = _spec_snapshot(spec), snapshot(graph)
    for k in osnap:
        assert_bitexact(esnap[k], osnap[k], '{} {}'.format(name, k))
    for r, s in zip(rels, S_o):
        assert_bitexact(npy(r.get_scale_vec()), s)


@pytest.mark.parametrize('merged,chain_first', [('0', '1'), ('1', '0'), ('0', '0')])
def test_launch_modes_give_identical_results(engine, monkeypatch, merged, chain_first):
    """One launch per sweep vs one per dependency level, longest-chain-first vs list order inside a level: the same
    kernel, the same tiles, only the launch slicing / table order differ -- results are bit-identical."""
    monkeypatch.setenv('DFQ_LE_MERGED', merged)
    monkeypatch.setenv('DFQ_LE_CHAIN_FIRST', chain_first)
    for name, seed in (('tiny_mobile', 0), ('tiny_res', 0)):
        gold = net_fixture(name, seed, '')
        model, graph, bottoms = _build(name, seed, gold, engine)
        spec = graphspec.from_torch(graph, bottoms, TARG)
        lt.merge_batchnorm(model, graph, bottoms, TARG)
        orc.merge_batchnorm(spec)
        rels = rel.create_relation(graph, bottoms, TARG)
        plan = dfq.build_le_plan(graph, rels, TARG)
        assert plan.levels == (1 if merged == '1' else plan.depth)
        res = plan.run()
        n_o, _ = orc.cross_layer_equalization(spec, orc.create_relation(spec))
        assert res['sweeps'] == n_o
        osnap, esnap = _spec_snapshot(spec), snapshot(graph)
        for k in osnap:
            assert_bitexact(esnap[k], osnap[k], '{} {} (merged={}, chain_first={})'.format(name, k, merged, chain_first))


@pytest.mark.parametrize('merged,tile_elems', [('1', None), ('0', None), ('1', 48)])
def test_depthwise_rows_take_their_own_statistics(engine, monkeypatch, merged, tile_elems):
    """Round 4 (streaming engine): a depthwise layer inside a chain is walked by one thread per row, so the thread takes the
    row range of t = fl(w / s_prev) itself, publishes it for its relation's column tiles (which wait for it inside the launch)
    and the previous relation's read-only pass over the layer is not launched (LeRelDev::local_r1).  Fewer workgroups, the same
    bits: against the oracle and against the plan that keeps the pass (DFQ_LE_LOCAL_R1=0), data-dependent sweep count included.
    (General tiles only, DFQ_LE_CF=0: with free-running segments the blocks of this network are not on these tiles at all.)"""
    monkeypatch.setenv('DFQ_LE_RESIDENT', '0')
    monkeypatch.setenv('DFQ_LE_CF', '0')
    monkeypatch.setenv('DFQ_LE_MERGED', merged)
    if tile_elems:
        monkeypatch.setenv('DFQ_LE_TILE_ELEMS', str(tile_elems))
    for name, seed, suffix in (('tiny_mobile', 0, ''), ('tiny_mobile', 2, '_signed')):
        gold = net_fixture(name, seed, suffix)
        signed = bool(gold['cfg'][1])
        out, groups = {}, {}
        for local in ('1', '0'):
            monkeypatch.setenv('DFQ_LE_LOCAL_R1', local)
            model, graph, bottoms = _build(name, seed, gold, engine)
            spec = graphspec.from_torch(graph, bottoms, TARG)
            lt.merge_batchnorm(model, graph, bottoms, TARG)
            orc.merge_batchnorm(spec)
            rels = rel.create_relation(graph, bottoms, TARG)
            plan = dfq.build_le_plan(graph, rels, TARG)
            groups[local] = sum(plan.level_info(l)['workgroups'] for l in range(plan.levels))
            res = plan.run(signed=signed)
            n_o, S_o = orc.cross_layer_equalization(spec, orc.create_relation(spec), signed=signed)
            assert res['sweeps'] == n_o
            osnap, esnap = _spec_snapshot(spec), snapshot(graph)
            for k in osnap:
                assert_bitexact(esnap[k], osnap[k], '{} {} local_r1={}'.format(name, k, local))
            for a, b in zip(plan.scale_cum, S_o):
                assert_bitexact(npy(a), b, 'cumulative S')
            out[local] = esnap
            plan.close()
        assert groups['1'] < groups['0'], groups          # the read-only passes over the depthwise layers are gone
        for k in out['1']:
            assert_bitexact(out['1'][k], out['0'][k], '{} {}'.format(name, k))


class _PointwiseChain(nn.Module):
    """conv1x1 chains with interior layers whose rows are 16-byte vectors: c0 -> c1 -> c2 -> c3 (two consecutive interior layers),
    and a second chain behind a ReLU6 with a grouped interior layer"""

    def __init__(self):
        super().__init__()
        self.c0 = nn.Conv2d(3, 16, 1)
        self.b0 = nn.BatchNorm2d(16)
        self.c1 = nn.Conv2d(16, 40, 1)
        self.b1 = nn.BatchNorm2d(40)
        self.c2 = nn.Conv2d(40, 24, 1)
        self.b2 = nn.BatchNorm2d(24)
        self.c3 = nn.Conv2d(24, 12, 1)
        self.b3 = nn.BatchNorm2d(12)
        self.r6 = nn.ReLU6()
        self.d0 = nn.Conv2d(12, 32, 1)
        self.e0 = nn.BatchNorm2d(32)
        self.d1 = nn.Conv2d(32, 32, 1, groups=4)          # rows of 8 floats, four groups
        self.e1 = nn.BatchNorm2d(32)
        self.d2 = nn.Conv2d(32, 6, 1)

    def forward(self, x):
        x = torch.relu(self.b0(self.c0(x)))
        x = torch.relu(self.b1(self.c1(x)))
        x = torch.relu(self.b2(self.c2(x)))
        x = self.r6(self.b3(self.c3(x)))
        x = torch.relu(self.e0(self.d0(x)))
        x = torch.relu(self.e1(self.d1(x)))
        return self.d2(x)


@pytest.mark.parametrize('merged', ['1', '0'])
@pytest.mark.parametrize('signed', [False, True])
def test_full_row_tiles_take_their_own_statistics(engine, monkeypatch, merged, signed):
    """Round 4 (streaming engine, opt-in DFQ_LE_LOCAL_ROW): an interior layer whose rows are 16-byte vectors is tiled in FULL rows; the
    tile takes the row ranges of t = fl(w / s_prev) itself (a pass over its registers, a reduction per row through LDS),
    publishes them for its relation's column tiles, and the previous relation's read-only pass over the layer is not
    launched.  Two consecutive such layers, a grouped one, several row blocks: bit-identical to the oracle and to the plan
    that keeps the passes (DFQ_LE_LOCAL_R1=0), data-dependent sweep count included."""
    from dfq_amd import fxgraph
    monkeypatch.setenv('DFQ_LE_RESIDENT', '0')
    monkeypatch.setenv('DFQ_LE_MERGED', merged)
    monkeypatch.setenv('DFQ_LE_LOCAL_ROW', '512')            # opt-in (measured no faster at batch 32: dfq_le.hip)
    out, groups = {}, {}
    for local in ('1', '0'):
        monkeypatch.setenv('DFQ_LE_LOCAL_R1', local)
        torch.manual_seed(5)
        model = _PointwiseChain().eval()
        gen = torch.Generator().manual_seed(9)
        synthetic.init_weights(model, gen)
        graph, bottoms = fxgraph.trace(model)
        spec = graphspec.from_torch(graph, bottoms, TARG)
        model.to(engine.device)
        lt.merge_batchnorm(model, graph, bottoms, TARG)
        orc.merge_batchnorm(spec)
        rels = rel.create_relation(graph, bottoms, TARG)
        assert len(rels) == 5
        plan = dfq.build_le_plan(graph, rels, TARG)
        groups[local] = sum(plan.level_info(l)['workgroups'] for l in range(plan.levels))
        res = plan.run(signed=signed)
        n_o, S_o = orc.cross_layer_equalization(spec, orc.create_relation(spec), signed=signed)
        assert res['sweeps'] == n_o
        osnap, esnap = _spec_snapshot(spec), snapshot(graph)
        for k in osnap:
            assert_bitexact(esnap[k], osnap[k], '{} local_r1={}'.format(k, local))
        for a, b in zip(plan.scale_cum, S_o):
            assert_bitexact(npy(a), b, 'cumulative S')
        out[local] = esnap
        plan.close()
    assert groups['1'] < groups['0'], groups
    for k in out['1']:
        assert_bitexact(out['1'][k], out['0'][k], k)


@pytest.mark.parametrize('background', [False, True])
def test_batched_plan_matches_separate_runs(engine, monkeypatch, background):
    """Several networks in one plan (every launch covers the batch): each network must end exactly
    where a plan of its own ends, including its own sweep count.  `background`: the lean launches of the free-running layers on
    the plan's second stream, two groups of look-ahead (DFQ_LE_CF_BG=1, dfq_le_cf.hpp) -- networks of one batch stop at
    different sweeps, i.e. inside different groups."""
    monkeypatch.delenv('DFQ_LE_CF_BG', raising=False)
    if background:
        monkeypatch.setenv('DFQ_LE_CF_BG', '1')
    cases = [('tiny_mobile', 0, ''), ('tiny_cat', 0, ''), ('tiny_mobile', 1, '_abs'), ('tiny_res', 0, '')]
    items, specs = [], []
    for name, seed, suffix in cases:
        gold = net_fixture(name, seed, suffix)
        model, graph, bottoms = _build(name, seed, gold, engine)
        spec = graphspec.from_torch(graph, bottoms, TARG)
        lt.merge_batchnorm(model, graph, bottoms, TARG)
        orc.merge_batchnorm(spec)
        items.append((graph, rel.create_relation(graph, bottoms, TARG)))
        specs.append(spec)
    plan = dfq.build_le_plan_batch(items, TARG)
    assert plan.n_nets == len(cases) and plan.lean_background == background and plan.free_running_group == 8
    plan.run()
    results, all_done = plan.query_all()
    assert all_done
    for (graph, rels), spec, res in zip(items, specs, results):
        n_o, S_o = orc.cross_layer_equalization(spec, orc.create_relation(spec))
        assert res['sweeps'] == n_o
        osnap, esnap = _spec_snapshot(spec), snapshot(graph)
        for k in osnap:
            assert_bitexact(esnap[k], osnap[k], k)
        for r, s in zip(rels, S_o):
            assert_bitexact(npy(r.get_scale_vec()), s)


@pytest.mark.parametrize('skew,one_launch', [('0', None), ('0.5', None), ('3', '1'), ('0.25', '0'), ('40', '1')])
def test_batched_bias_correction_matches_separate_runs(engine, monkeypatch, skew, one_launch):
    """The j-th correction steps of all networks of a batch share one launch; every network must get
    exactly what its own plan gives it.  `skew` (DFQ_BC_SKEW): the networks of the batch run that many chain positions behind
    each other in the launch's workgroup order (dfq_bc.hip: some stream their large layers while others hand over) -- the order
    of the workgroups changes, no result does; with and without the min/max blocks woven into the launch."""
    monkeypatch.setenv('DFQ_BC_SKEW', skew)
    monkeypatch.delenv('DFQ_BC_ONE_LAUNCH', raising=False)
    if one_launch is not None:
        monkeypatch.setenv('DFQ_BC_ONE_LAUNCH', one_launch)
    cases = [('tiny_mobile', 0, ''), ('tiny_cat', 0, ''), ('tiny_res', 0, '')]
    items, singles = [], []
    for name, seed, suffix in cases:
        gold = net_fixture(name, seed, suffix)
        for target in (items, singles):
            model, graph, bottoms = _build(name, seed, gold, engine)
            lt.merge_batchnorm(model, graph, bottoms, TARG)
            load_stage(graph, gold, 'abs')
            target.append((graph, bottoms))
    plan = dfq.build_bc_plan_batch(items, TARG)
    plan.run()
    for (graph, bottoms), (g1, b1), (name, seed, suffix) in zip(items, singles, cases):
        dfq.bias_correction(g1, b1, TARG)
        a, b = snapshot(graph), snapshot(g1)
        for k in b:
            assert_bitexact(a[k], b[k], '{} {}'.format(name, k))
        compare_stage(a, net_fixture(name, seed, suffix), 'bc', what=name)


@pytest.mark.parametrize('mode', ['tagged', 'counters', 'per-position'])
def test_bias_correction_hand_over_protocols_agree(engine, monkeypatch, mode):
    """The one-launch chain hands beta~ / the ReLU moment from step to step as tagged 64-bit values (default) or behind
    per-step counters (DFQ_BC_TAGGED=0); DFQ_BC_MERGED=0 makes every chain position its own launch.  Same arithmetic:
    identical results, also when the plan is run a second time on fresh inputs (the slots then carry the next epoch)."""
    for k in ('DFQ_BC_TAGGED', 'DFQ_BC_MERGED'):
        monkeypatch.delenv(k, raising=False)
    if mode == 'counters':
        monkeypatch.setenv('DFQ_BC_TAGGED', '0')
    if mode == 'per-position':
        monkeypatch.setenv('DFQ_BC_MERGED', '0')
    for name, seed, suffix in [('tiny_mobile', 0, ''), ('tiny_cat', 0, ''), ('tiny_res', 0, '')]:
        gold = net_fixture(name, seed, suffix)
        model, graph, bottoms = _build(name, seed, gold, engine)
        lt.merge_batchnorm(model, graph, bottoms, TARG)
        load_stage(graph, gold, 'abs')
        start = {k: v.copy() for k, v in snapshot(graph).items()}
        plan, _ = dfq.build_bc_plan(graph, bottoms, TARG)
        assert plan.tagged == (mode == 'tagged') and not plan.last_run_tagged
        assert not plan.one_launch                                # (a single network keeps min/max as its own launch: measured slower woven in)
        plan.run()
        # the run itself, on the stream the tests run on -- the NULL stream, which until round 5 was mistaken for the plan's
        # (not yet created) graph-recording stream and got the counters
        assert plan.last_run_tagged == (mode == 'tagged')
        first = snapshot(graph)
        compare_stage(first, gold, 'bc', what='{} {}'.format(name, mode))
        # second run of the SAME plan from the same start: bit-identical to the first
        load_stage(graph, gold, 'abs')
        for k, v in snapshot(graph).items():
            assert_bitexact(v, start[k], 'reload {}'.format(k))
        plan.run()
        for k, v in snapshot(graph).items():
            assert_bitexact(v, first[k], '{} second run {} ({})'.format(name, k, mode))
        plan.close()


@pytest.mark.parametrize('mode', ['tagged', 'counters', 'per-position'])
def test_folded_depthwise_steps_are_invisible(engine, monkeypatch, mode):
    """Round 4: a depthwise correction step (one input per group: channel o needs E[o] only) is performed by the per-row tail
    of the step that produces its source BN (dfq_bc.hip: BcFoldDev) -- one hand-over through the memory system less per
    depthwise layer.  Same operations in the same order: corrected biases, BN proxies and correction vectors are
    BIT-IDENTICAL to the plan that keeps every step (DFQ_BC_FOLD=0), under all three hand-over protocols, and a second run
    of the plan repeats the first."""
    for k in ('DFQ_BC_TAGGED', 'DFQ_BC_MERGED', 'DFQ_BC_FOLD'):
        monkeypatch.delenv(k, raising=False)
    if mode == 'counters':
        monkeypatch.setenv('DFQ_BC_TAGGED', '0')
    if mode == 'per-position':
        monkeypatch.setenv('DFQ_BC_MERGED', '0')
    seen_fold = 0
    for name, seed, suffix in [('tiny_mobile', 0, ''), ('tiny_mobile', 2, '_signed'), ('tiny_cat', 0, ''), ('tiny_res', 0, '')]:
        gold = net_fixture(name, seed, suffix)
        signed = suffix == '_signed'
        results = {}
        for fold in ('1', '0'):
            monkeypatch.setenv('DFQ_BC_FOLD', fold)
            model, graph, bottoms = _build(name, seed, gold, engine)
            lt.merge_batchnorm(model, graph, bottoms, TARG)
            load_stage(graph, gold, 'abs')
            plan, keys = dfq.build_bc_plan(graph, bottoms, TARG)
            n_steps = len(keys)
            if fold == '0':
                assert plan.folded_steps == 0 and plan.chain_steps == n_steps
            else:
                assert plan.chain_steps == n_steps - plan.folded_steps
                seen_fold += plan.folded_steps
            plan.run(signed=signed)
            snap = snapshot(graph)
            corr = [npy(plan.correction(i)) for i in range(n_steps)]
            load_stage(graph, gold, 'abs')
            plan.run(signed=signed)                              # second run of the same plan
            for k, v in snapshot(graph).items():
                assert_bitexact(v, snap[k], '{} second run {}'.format(name, k))
            plan.close()
            results[fold] = (snap, corr)
            compare_stage(snap, gold, 'bc', what='{} fold={} {}'.format(name, fold, mode))
        for k in results['1'][0]:
            assert_bitexact(results['1'][0][k], results['0'][0][k], '{} folded vs unfolded {}'.format(name, k))
        for i, (a, b) in enumerate(zip(results['1'][1], results['0'][1])):
            assert_bitexact(a, b, '{} correction vector of step {}'.format(name, i))
    assert seen_fold > 0, 'no fixture exercised a folded step'


def test_bias_correction_intermediates_against_oracle(engine, monkeypatch):
    """eps (quant-error row sums, dfq.py:216-219) and the correction vectors (dfq.py:281-287) read back from
    the plan: eps is float32 elementwise work in the oracle's order (bit-exact), the matvec is 1e-5.  The chain forms the
    row sums in registers; DFQ_BC_EPS=1 at plan creation makes one more launch materialise them (same device function)
    for this inspection, and the corrected biases must not depend on that switch."""
    monkeypatch.setenv('DFQ_BC_EPS', '1')
    gold = net_fixture('tiny_res', 0, '')
    model, graph, bottoms = _build('tiny_res', 0, gold, engine)
    lt.merge_batchnorm(model, graph, bottoms, TARG)
    load_stage(graph, gold, 'abs')
    spec = graphspec.from_torch(graph, bottoms, TARG)
    collect = {}
    orc.bias_correction(spec, collect=collect)
    plan, keys = dfq.build_bc_plan(graph, bottoms, TARG)
    plan.run()
    assert keys == list(collect.keys())
    for step, k in enumerate(keys):
        assert_bitexact(npy(plan.eps(step)), collect[k]['eps'].reshape(npy(plan.eps(step)).shape), 'eps of {}'.format(k))
        assert_close(npy(plan.correction(step)), collect[k]['bias'].reshape(-1), 'correction of {}'.format(k))
    assert plan.weight_elements == sum(spec.nodes[k].weight.size for k in keys)
    with_debug = snapshot(graph)
    plan.close()
    monkeypatch.delenv('DFQ_BC_EPS')
    load_stage(graph, gold, 'abs')
    plain, _ = dfq.build_bc_plan(graph, bottoms, TARG)
    plain.run()
    with pytest.raises(RuntimeError, match='DFQ_BC_EPS'):
        plain.eps(0)
    for k, v in snapshot(graph).items():
        assert_bitexact(v, with_debug[k], 'with / without the materialised row sums: {}'.format(k))
    plain.close()


def test_graph_replay_mode(engine, monkeypatch):
    """DFQ_GRAPH=1: the sweep run and the BC chain are recorded once and replayed as hipGraphs."""
    monkeypatch.setenv('DFQ_GRAPH', '1')
    gold = net_fixture('tiny_cat', 3, '_abs')
    model, graph, bottoms = _build('tiny_cat', 3, gold, engine)
    spec = graphspec.from_torch(graph, bottoms, TARG)
    lt.merge_batchnorm(model, graph, bottoms, TARG)
    orc.merge_batchnorm(spec)
    rels = rel.create_relation(graph, bottoms, TARG)
    dfq.cross_layer_equalization(graph, rels, TARG)
    n_o, _ = orc.cross_layer_equalization(spec, orc.create_relation(spec))
    assert dfq.last_equalization['sweeps'] == n_o
    dfq.bias_correction(graph, bottoms, TARG)
    orc.bias_correction(spec)
    osnap, esnap = _spec_snapshot(spec), snapshot(graph)
    for k in osnap:
        if k.endswith('.w') or k.endswith('.fw'):
            assert_bitexact(esnap[k], osnap[k], k)
        else:
            assert_close(esnap[k], osnap[k], k)


def test_max_sweeps_and_restart(engine):
    gold = net_fixture('tiny_mobile', 0, '')
    model, graph, bottoms = _build('tiny_mobile', 0, gold, engine)
    lt.merge_batchnorm(model, graph, bottoms, TARG)
    rels = rel.create_relation(graph, bottoms, TARG)
    spec = graphspec.from_torch(graph, bottoms, TARG)
    dfq.cross_layer_equalization(graph, rels, TARG, max_sweeps=3)
    assert dfq.last_equalization['sweeps'] == 3
    orels = orc.create_relation(spec)
    n, S = orc.cross_layer_equalization(spec, orels, max_sweeps=3)
    osnap, esnap = _spec_snapshot(spec), snapshot(graph)
    for k in osnap:
        assert_bitexact(esnap[k], osnap[k], k)
    # a second call continues from the current weights and accumulates S (relation.py:20-24)
    dfq.cross_layer_equalization(graph, rels, TARG, max_sweeps=2)
    n, S2 = orc.cross_layer_equalization(spec, orels, max_sweeps=2)
    for r, a, b in zip(rels, S, S2):      # ((s1*s2*s3)*s4)*s5 vs (s1*s2*s3)*(s4*s5): equal up to rounding
        assert_close(npy(r.get_scale_vec()), (a * b).astype(F32), tol=1e-6)
    osnap, esnap = _spec_snapshot(spec), snapshot(graph)
    for k in osnap:
        assert_bitexact(esnap[k], osnap[k], k)


def test_clip_weight(engine):
    gold = net_fixture('tiny_cat', 0, '')
    model, graph, bottoms = _build('tiny_cat', 0, gold, engine)
    before = snapshot(graph)
    dfq.clip_weight(graph, [-0.05, 0.07], TARG)
    after = snapshot(graph)
    for k in before:
        if k.endswith('.w'):
            assert_bitexact(after[k], np.clip(before[k], F32(-0.05), F32(0.07)))


def _spec_snapshot(spec):
    snap = {}
    for i, k in enumerate(spec.order):
        n = spec.nodes[k]
        if n.kind == 'targ':
            snap['L{}.w'.format(i)] = n.weight
            if n.bias is not None:
                snap['L{}.b'.format(i)] = n.bias
        elif n.kind == 'bn' and n.fake_weight is not None:
            snap['L{}.fw'.format(i)] = n.fake_weight
            snap['L{}.fb'.format(i)] = n.fake_bias
    return snap


def test_geometry_corner_cases_against_oracle(engine):
    """tiny_wide: 3-input stem, grouped 3x3, depthwise 5x5, a 70-input and a 1600-input row -- every work
    split of the equalisation tiles and of the bias-correction matvec (narrow rows sharing a wave,
    multi-slot rows, rows longer than the register preload).  LE bit-exact, BC within 1e-5."""
    model, graph, bottoms = synthetic.build('tiny_wide', seed=3)
    spec = graphspec.from_torch(graph, bottoms, TARG)
    model.to(engine.device)
    lt.merge_batchnorm(model, graph, bottoms, TARG)
    orc.merge_batchnorm(spec)
    rels = rel.create_relation(graph, bottoms, TARG)
    orels = orc.create_relation(spec)
    dfq.cross_layer_equalization(graph, rels, TARG, max_sweeps=6)
    n_o, S_o = orc.cross_layer_equalization(spec, orels, max_sweeps=6)
    assert dfq.last_equalization['sweeps'] == n_o == 6
    osnap, esnap = _spec_snapshot(spec), snapshot(graph)
    for k in osnap:
        assert_bitexact(esnap[k], osnap[k], 'tiny_wide LE {}'.format(k))
    dfq.bias_correction(graph, bottoms, TARG)
    orc.bias_correction(spec)
    osnap, esnap = _spec_snapshot(spec), snapshot(graph)
    for k in osnap:
        assert_close(esnap[k], osnap[k], 'tiny_wide BC {}'.format(k))


def test_host_resident_model_is_staged_and_written_back(engine):
    """The reference's default flow keeps the model on the CPU: the package shadows every tensor on the device,
    runs the passes there and writes the results back into the caller's CPU tensors (same objects)."""
    gold = net_fixture('tiny_mobile', 0, '')
    model, graph, bottoms = _build('tiny_mobile', 0, gold, Engine_cpu())
    spec = graphspec.from_torch(graph, bottoms, TARG)
    ids = {k: id(m.weight) for k, m in graph.items() if type(m) in TARG}
    lt.merge_batchnorm(model, graph, bottoms, TARG)
    orc.merge_batchnorm(spec)
    rels = rel.create_relation(graph, bottoms, TARG)
    dfq.cross_layer_equalization(graph, rels, TARG)
    n_o, S_o = orc.cross_layer_equalization(spec, orc.create_relation(spec))
    assert dfq.last_equalization['sweeps'] == n_o
    dfq.bias_correction(graph, bottoms, TARG)
    orc.bias_correction(spec)
    osnap, esnap = _spec_snapshot(spec), snapshot(graph)
    for k in osnap:
        assert_close(esnap[k], osnap[k], 'host-resident {}'.format(k))
    for k, m in graph.items():
        if type(m) in TARG:
            assert m.weight.device.type == 'cpu' and id(m.weight) == ids[k]
    for r, s in zip(rels, S_o):
        assert r.get_scale_vec().device.type == 'cpu'
        assert_bitexact(npy(r.get_scale_vec()), s)


def test_packed_staging_of_foreign_tensors(engine):
    """Tensors the kernels cannot use in place (here: a float64 model on the host -- on the GPU engine also simply a host
    model) are shadowed through ONE packed copy each way (`Stage.prefetch`): same results as the float32 model, written back
    into the caller's own tensors in their own dtype, weights of different layers not overlapping in the packed buffer."""
    from dfq_amd import _ffi
    gold = net_fixture('tiny_mobile', 0, '')
    model32, graph32, bottoms32 = _build('tiny_mobile', 0, gold, Engine_cpu())
    model64, graph64, bottoms64 = _build('tiny_mobile', 0, gold, Engine_cpu())
    model64.double()
    res = []
    for model, graph, bottoms in ((model32, graph32, bottoms32), (model64, graph64, bottoms64)):
        ids = {k: id(m.weight) for k, m in graph.items() if type(m) in TARG}
        lt.merge_batchnorm(model, graph, bottoms, TARG)
        rels = rel.create_relation(graph, bottoms, TARG)
        dfq.cross_layer_equalization(graph, rels, TARG)
        dfq.bias_correction(graph, bottoms, TARG)
        lt.quantize_targ_layer(graph, 8, 16, TARG)
        for k, m in graph.items():
            if type(m) in TARG:
                assert id(m.weight) == ids[k] and m.weight.device.type == 'cpu'
        res.append((dfq.last_equalization['sweeps'], snapshot(graph)))
    assert all(m.weight.dtype == torch.float64 for m in graph64.values() if type(m) in TARG)
    assert res[0][0] == res[1][0]
    for k in res[0][1]:
        assert_bitexact(np.asarray(res[1][1][k], dtype=F32), res[0][1][k], 'float64 host model: {}'.format(k))
    # the packing itself: one flat buffer, 256-byte aligned segments, views of the right shape, one copy back
    st = _ffi.Stage()
    a, b, c = torch.arange(5, dtype=torch.float64), torch.ones(3, 7, dtype=torch.float64)[:, ::2], torch.zeros(0, dtype=torch.float64)
    st.prefetch([a, b, None, c, a])
    assert len(st._packs) == 1 and len(st._packs[0][1]) == 2
    da, db = st.bind(a), st.bind(b)
    assert da.dtype == torch.float32 and tuple(db.shape) == (3, 4) and db.is_contiguous()
    assert (db.data_ptr() - da.data_ptr()) == 4 * _ffi.Stage._ALIGN
    da += 1
    db *= 3
    st.writeback()
    assert a.tolist() == [1, 2, 3, 4, 5] and b.tolist() == [[3.0] * 4] * 3


@pytest.mark.gpu
def test_one_launch_sweep_is_stable_against_per_level_launches(monkeypatch):
    """Stress for the in-launch dependency protocol (device-scope atomics + sc1 loads across XCDs): a batch of
    full-size MobileNetV2 run several times with one launch per sweep must equal, bit for bit and every time, the
    same batch run with one launch per dependency level (where every dependency is a kernel boundary)."""
    dev = torch.device('cuda', 0)

    def run(merged):
        monkeypatch.setenv('DFQ_LE_MERGED', merged)
        items = []
        for seed in range(4):
            model, graph, bottoms = synthetic.build('mobilenet_v2', seed=seed)
            model.to(dev)
            lt.merge_batchnorm(model, graph, bottoms, TARG)
            items.append((graph, rel.create_relation(graph, bottoms, TARG)))
        plan = dfq.build_le_plan_batch(items, TARG)
        plan.run()
        res, _ = plan.query_all()
        return [r['sweeps'] for r in res], [snapshot(g) for g, _ in items]
    sweeps_ref, snaps_ref = run('0')
    for rep in range(3):
        sweeps, snaps = run('1')
        assert sweeps == sweeps_ref
        for a, b in zip(snaps, snaps_ref):
            for k in b:
                assert_bitexact(a[k], b[k], 'repetition {} {}'.format(rep, k))


@pytest.mark.gpu
@pytest.mark.parametrize('persist', ['0', '1'])
def test_two_host_threads_two_streams(monkeypatch, persist):
    """Two host threads feed two HIP streams with batched equalisation + bias correction at full size, as bench.py does.
    Launches with in-launch waits from different streams must never overlap (SpinGuard holds its mutex from the stream
    wait to the event record); with DFQ_LE_PERSIST=1 every sweep launch also needs all its workgroups resident.
    Every network must end exactly where the same batch ends on one stream."""
    import threading
    dev = torch.device('cuda', 0)
    monkeypatch.setenv('DFQ_LE_PERSIST', persist)

    def make(seed0):
        items = []
        for seed in range(seed0, seed0 + 8):
            model, graph, bottoms = synthetic.build('mobilenet_v2', seed=seed)
            model.to(dev)
            lt.merge_batchnorm(model, graph, bottoms, TARG)
            items.append((graph, bottoms, rel.create_relation(graph, bottoms, TARG)))
        le = dfq.build_le_plan_batch([(g, r) for g, _, r in items], TARG)
        bc = dfq.build_bc_plan_batch([(g, b) for g, b, _ in items], TARG)
        assert (le.sweep_workgroups > 0) == (persist == '1')
        return items, le, bc

    def work(unit):
        items, le, bc = unit
        le.enqueue(0, restart=True)
        le.enqueue(47, restart=False)
        bc.run()

    ref_units = [make(0), make(8)]
    for u in ref_units:
        work(u)
    torch.cuda.synchronize()
    ref = [([r['sweeps'] for r in u[1].query_all()[0]], [snapshot(g) for g, _, _ in u[0]]) for u in ref_units]
    assert max(ref[0][0]) == 47 and min(ref[0][0]) >= 40

    for rep in range(2):
        units = [make(0), make(8), make(0), make(8)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        errors = []

        def worker(i):
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(streams[i]):
                    for u in units[i::2]:
                        work(u)
            except Exception as e:      # surfaced below: an exception in a thread would otherwise pass silently
                errors.append(e)
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        assert not errors, errors
        for j, u in enumerate(units):
            sweeps_ref, snaps_ref = ref[j % 2]
            with torch.cuda.stream(streams[j % 2]):
                assert [r['sweeps'] for r in u[1].query_all()[0]] == sweeps_ref      # also surfaces an abandoned wait
                u[2].status()
            for (g, _, _), b in zip(u[0], snaps_ref):
                a = snapshot(g)
                for k in b:
                    assert_bitexact(a[k], b[k], 'repetition {} unit {} {}'.format(rep, j, k))


@pytest.mark.gpu
def test_heterogeneous_full_size_batch_matches_single_plans():
    """One batched plan over DIFFERENT architectures at full size (MobileNetV2, ResNet-18, a second MobileNetV2
    with other weights): every network must end bit-identical to a plan of its own -- weights, cumulative scales,
    sweep count -- and its bias correction (batched BC plan, networks with different layer counts per launch)
    must equal the single-network one."""
    dev = torch.device('cuda', 0)
    cases = [('mobilenet_v2', 0), ('resnet18', 0), ('mobilenet_v2', 5)]
    batch, single = [], []
    for name, seed in cases:
        for target in (batch, single):
            model, graph, bottoms = synthetic.build(name, seed=seed)
            model.to(dev)
            lt.merge_batchnorm(model, graph, bottoms, TARG)
            target.append((model, graph, bottoms, rel.create_relation(graph, bottoms, TARG)))
    plan = dfq.build_le_plan_batch([(g, r) for (_, g, _, r) in batch], TARG)
    plan.run()
    results, _ = plan.query_all()
    bc = dfq.build_bc_plan_batch([(g, b) for (_, g, b, _) in batch], TARG)
    bc.run()
    first_rel = 0
    for (name, seed), (_, gb, _, rb), (_, gs, bs, rs), res in zip(cases, batch, single, results):
        dfq.cross_layer_equalization(gs, rs, TARG)
        assert res['sweeps'] == dfq.last_equalization['sweeps'], name
        for i, r2 in enumerate(rs):            # the plan keeps the cumulative S of all networks' relations in list order
            assert_bitexact(npy(plan.scale_cum[first_rel + i]), npy(r2.get_scale_vec()), '{} cumulative S'.format(name))
        first_rel += len(rs)
        dfq.bias_correction(gs, bs, TARG)
        a, b = snapshot(gb), snapshot(gs)
        for k in b:
            assert_bitexact(a[k], b[k], '{} s{} {}'.format(name, seed, k))


# ---------------------------------------------------------------------------------------------
# BASELINE.json configurations at full size (GPU only: the emulation would take minutes)
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('net,max_sweeps,pinned', [('mobilenet_v2', None, False), ('resnet18', None, False), ('deeplab_mnv2', 12, True),
                                                   ('deeplab_mnv2', None, False), ('deeplab_mnv2', 60, True)])
def test_full_size_networks_against_oracle(net, max_sweeps, pinned):
    """configs[1..3] of BASELINE.json: the whole LE + BC + quantise pass on the real layer shapes.
    LE must be bit-identical to the oracle (sweep count included), BC within 1e-5, int8 codes of the
    weights bit-identical.  DeepLab three ways: 12 pinned sweeps, the data-dependent loop (46 sweeps: on the
    reference's 35-relation graph the reference's own loop terminates there, tests/golden/fullconv_deeplab_mnv2_s0.npz)
    and 60 PINNED sweeps -- the configuration bench.py times (`config.others`, `sharded`)."""
    pin = dict(converge_thres=-1.0, converge_count=10 ** 9) if pinned else {}
    dev = torch.device('cuda', 0)
    model, graph, bottoms = synthetic.build(net, seed=0)
    spec = graphspec.from_torch(graph, bottoms, TARG)
    model.to(dev)
    lt.merge_batchnorm(model, graph, bottoms, TARG)
    orc.merge_batchnorm(spec)
    rels = rel.create_relation(graph, bottoms, TARG)
    orels = orc.create_relation(spec)
    keys = list(graph.keys())
    assert [[keys.index(k) for k in r.get_idxs()] for r in rels] == [[spec.order.index(k) for k in r] for r in orels]

    dfq.cross_layer_equalization(graph, rels, TARG, max_sweeps=max_sweeps, **pin)
    n_o, S_o = orc.cross_layer_equalization(spec, orels, max_sweeps=max_sweeps, **pin)
    assert dfq.last_equalization['sweeps'] == n_o == (max_sweeps if pinned else n_o)
    if net == 'deeplab_mnv2' and not pinned:
        assert n_o == 46
    osnap, esnap = _spec_snapshot(spec), snapshot(graph)
    for k in osnap:
        assert_bitexact(esnap[k], osnap[k], '{} LE {}'.format(net, k))
    for r, s in zip(rels, S_o):
        assert_bitexact(npy(r.get_scale_vec()), s, 'cumulative S')
    # property of the converged state: paired ranges are equal (README "equalization")
    if max_sweeps is None:
        for r in rels:
            w1 = npy(graph[r.get_idxs()[0]].weight)
            w2 = npy(graph[r.get_idxs()[1]].weight)
            g = w1.shape[0] // w2.shape[1] if w1.shape[0] != w2.shape[1] else 1
            r1 = w1.reshape(w1.shape[0], -1).max(1) - w1.reshape(w1.shape[0], -1).min(1)
            cols = w2.reshape(g, w2.shape[0] // g, w2.shape[1], -1).transpose(0, 2, 1, 3).reshape(w1.shape[0], -1)
            r2 = cols.max(1) - cols.min(1)
            live = (r1 > 1e-6) & (r2 > 1e-6)
            np.testing.assert_allclose(r1[live], r2[live], rtol=2e-3)

    dfq.bias_correction(graph, bottoms, TARG)
    orc.bias_correction(spec)
    osnap, esnap = _spec_snapshot(spec), snapshot(graph)
    for k in osnap:
        assert_close(esnap[k], osnap[k], '{} BC {}'.format(net, k))

    _, codes = lt.quantize_targ_layer(graph, 8, 16, TARG, return_codes=True)
    ocodes = orc.quantize_targ_layer(spec, 8, 16, return_codes=True)
    for k in codes:
        assert np.array_equal(codes[k].cpu().numpy(), ocodes[k].astype(np.int32)), 'int8 codes of {}'.format(k)
        assert len(np.unique(npy(graph[k].weight))) <= 256


@pytest.mark.gpu
@pytest.mark.parametrize('net,max_sweeps,pinned', [('mobilenet_v2', None, False), ('deeplab_mnv2', 12, True)])
def test_full_size_single_network_on_the_streaming_engine(monkeypatch, net, max_sweeps, pinned):
    """ONE full-size network on the streaming engine with free-running segments (DFQ_LE_CF_GROUP=4: by default a single network
    of this size keeps the general tiles, measured faster -- cf_group_from_env).  Its chains expand -> depthwise -> project are
    segments like the batch's, with column tiles as wide as the layer (until late in round 6 the plan refused every chain end
    whose column tile spans more than 127 channels: a bound meant for grouped layers' factor tables), and the result is the
    oracle's, bit for bit, sweep count included."""
    monkeypatch.setenv('DFQ_LE_RESIDENT', '0')
    monkeypatch.setenv('DFQ_LE_CF_GROUP', '4')
    pin = dict(converge_thres=-1.0, converge_count=10 ** 9) if pinned else {}
    dev = torch.device('cuda', 0)
    model, graph, bottoms = synthetic.build(net, seed=0)
    spec = graphspec.from_torch(graph, bottoms, TARG)
    model.to(dev)
    lt.merge_batchnorm(model, graph, bottoms, TARG)
    orc.merge_batchnorm(spec)
    rels = rel.create_relation(graph, bottoms, TARG)
    orels = orc.create_relation(spec)
    plan = dfq.build_le_plan(graph, rels, TARG)
    assert plan.resident_tiles == 0 and plan.free_running_elements > 1000000 and plan.free_running_group == 4
    out = plan.run(max_sweeps=max_sweeps, **pin)
    n_o, S_o = orc.cross_layer_equalization(spec, orels, max_sweeps=max_sweeps, **pin)
    assert out['sweeps'] == n_o
    osnap, esnap = _spec_snapshot(spec), snapshot(graph)
    for k in osnap:
        assert_bitexact(esnap[k], osnap[k], '{} LE {}'.format(net, k))
    for a, s in zip(plan.scale_cum, S_o):
        assert_bitexact(npy(a), s, 'cumulative S')
    plan.close()


# ---------------------------------------------------------------------------------------------
# the two equalisation engines: register-resident whole-loop launch vs streaming one-launch-per-sweep
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name,seed,signed', [('tiny_mobile', 0, False), ('tiny_res', 0, False), ('tiny_cat', 3, False), ('tiny_mobile', 2, True)])
def test_resident_and_streaming_engines_agree(engine, monkeypatch, name, seed, signed):
    """A single network runs the whole loop as ONE persistent launch with its weights in LDS
    (dfq_le_resident.hip); DFQ_LE_RESIDENT=0 forces the streaming kernel.  Same IEEE operations -> the weights, the
    [O] vectors, the cumulative scales, the sweep count and the loop state must be identical bit for bit, and both
    equal the oracle."""
    out = []
    for le_engine in LE_ENGINES:
        _select_le_engine(monkeypatch, le_engine)
        model, graph, bottoms = synthetic.build(name, seed=seed)
        model.to(engine.device)
        lt.merge_batchnorm(model, graph, bottoms, TARG)
        rels = rel.create_relation(graph, bottoms, TARG)
        plan = dfq.build_le_plan(graph, rels, TARG)
        assert (plan.resident_tiles > 0) == le_engine.startswith('resident'), plan.resident_reason
        if le_engine == 'streaming-persistent':
            assert plan.sweep_workgroups > 3
        elif le_engine == 'streaming-persistent-3wg':
            assert plan.sweep_workgroups == 3 and plan.level_info(0)['workgroups'] > 3
        else:
            assert plan.sweep_workgroups == 0
        res = plan.run(signed=signed)
        plan.stage.writeback()
        out.append((res, snapshot(graph), [npy(s) for s in plan.scale_cum]))
        plan.close()
    ra, sa, ca = out[0]
    for le_engine, (rb, sb, cb) in zip(LE_ENGINES[1:], out[1:]):
        assert ra == rb, '{}: loop state differs: {} vs {}'.format(le_engine, ra, rb)
        for k in sa:
            assert_bitexact(sa[k], sb[k], '{} {} {}'.format(le_engine, name, k))
        for a, b in zip(ca, cb):
            assert_bitexact(a, b, '{}: cumulative S'.format(le_engine))


@pytest.mark.parametrize('spec,ckpt,name,seed,signed',
                         [(sp, ck, 'tiny_mobile', 0, False) for sp, ck in [('0', '4'), ('1', '1'), ('2', '2'), ('2', '4'), ('3', '3'), ('4', '8'), ('6', '6')]] +
                         [(sp, ck, 'tiny_res', 0, False) for sp, ck in [('1', '1'), ('2', '4'), ('6', '6')]] +
                         [(sp, ck, 'tiny_cat', 3, True) for sp, ck in [('0', '4'), ('2', '2'), ('4', '8')]])
def test_speculation_past_the_verdict_is_invisible(engine, monkeypatch, name, seed, signed, spec, ckpt):
    """Round 4 (dfq_le_resident.hip, "speculation past the verdict"): the resident launch applies every sweep to its LDS
    tiles at once and may run DFQ_RES_SPEC sweeps ahead of the reducer's verdicts; when the loop stops it restores the
    newest checkpoint (every DFQ_RES_CKPT sweeps; checkpoint 0 = the untouched tensors) and replays the logged factors.
    Whatever the depth and the period -- no speculation at all, a checkpoint every sweep (no replay), a stop before the first
    checkpoint, a replay across several sweeps -- and however the sweeps are cut into launches, weights, [O] vectors,
    cumulative scales, sweep count and loop state are those of the oracle's sequential loop, bit for bit."""
    _select_le_engine(monkeypatch, 'resident')
    monkeypatch.setenv('DFQ_RES_SPEC', spec)
    monkeypatch.setenv('DFQ_RES_CKPT', ckpt)
    model, graph, bottoms = synthetic.build(name, seed=seed)
    model.to(engine.device)
    lt.merge_batchnorm(model, graph, bottoms, TARG)
    rels = rel.create_relation(graph, bottoms, TARG)
    spec0 = graphspec.from_torch(graph, bottoms, TARG)
    orels = orc.create_relation(spec0)
    # (1) the data-dependent loop in ONE launch
    plan = dfq.build_le_plan(graph, rels, TARG)
    assert plan.resident_tiles > 0, plan.resident_reason
    res = plan.run(signed=signed)
    st = plan.resident_stats()
    assert st['spec'] == int(spec) and st['ckpt'] == max(int(ckpt), int(spec), 1)
    assert st['max_undone'] <= int(spec), st
    if spec == '0':
        assert st['tiles_rolled_back'] == 0, st
    plan.stage.writeback()
    sp = spec0.clone()
    n_o, S_o = orc.cross_layer_equalization(sp, orels, signed=signed)
    assert res['sweeps'] == n_o, (res, n_o)
    snap = snapshot(graph)
    for i, k in enumerate(graph):
        n = sp.nodes[k]
        if n.kind == 'targ':
            assert_bitexact(snap['L{}.w'.format(i)], n.weight, 'w {}'.format(k))
            if n.bias is not None and 'L{}.b'.format(i) in snap:
                assert_bitexact(snap['L{}.b'.format(i)], n.bias, 'b {}'.format(k))
        elif n.kind == 'bn' and n.fake_weight is not None:
            assert_bitexact(snap['L{}.fw'.format(i)], n.fake_weight, 'fw {}'.format(k))
            assert_bitexact(snap['L{}.fb'.format(i)], n.fake_bias, 'fb {}'.format(k))
    for a, b in zip(plan.scale_cum, S_o):
        assert_bitexact(npy(a), b, 'cumulative S')
    plan.close()
    # (2) the same loop cut into launches of 1, 2, 3, ... sweeps (each launch reloads the tensors the previous one stored; the
    #     stop may fall anywhere inside a launch): after every launch the state is the oracle's after that many sweeps
    model, graph, bottoms = synthetic.build(name, seed=seed)
    model.to(engine.device)
    lt.merge_batchnorm(model, graph, bottoms, TARG)
    rels = rel.create_relation(graph, bottoms, TARG)
    plan = dfq.build_le_plan(graph, rels, TARG)
    plan.enqueue(0, restart=True, signed=signed)
    total, undone = 0, 0
    for n in (1, 2, 3, 1, 5, 1000):
        plan.enqueue(n, restart=False, signed=signed)
        r = plan.query()
        undone += plan.resident_stats()['sweeps_undone']
        total = min(total + n, n_o)
        assert r['sweeps'] == total, (r, total)
        sp = spec0.clone()
        orc.cross_layer_equalization(sp, orels, signed=signed, max_sweeps=total, converge_thres=-1.0, converge_count=10 ** 9)
        plan.stage.writeback()
        snap = snapshot(graph)
        for i, k in enumerate(graph):
            n_ = sp.nodes[k]
            if n_.kind == 'targ':
                assert_bitexact(snap['L{}.w'.format(i)], n_.weight, 'after {} sweeps: w {}'.format(total, k))
        if r['done']:
            break
    assert r['done'] and total == n_o
    plan.close()


# (the persistent-workgroup variant is slow on the CPU emulation: it runs at the default depth of batched plans only)
# Every (depth, engine) pair on tiny_mobile -- the network with free-running segments AND deferred one-way layers; a cross-section on
# the others (the whole product was 52 cases and a third of the CPU suite's time).
_DEFER_ENGINES = [('1', 'streaming-general'), ('2', 'streaming-general'), ('4', 'streaming-general'), ('4', 'streaming-persistent-3wg'),
                  ('1', 'streaming'), ('4', 'streaming'), ('2', 'streaming-cf2'), ('4', 'streaming-cf8'),
                  ('4', 'streaming-bg2'), ('1', 'streaming-bg4'), ('4', 'streaming-bg8'),
                  ('1', 'streaming-fused'), ('4', 'streaming-fused')]
_DEFER_CASES = ([(d, e, 'tiny_mobile', 0, False) for d, e in _DEFER_ENGINES if (d, e) not in (('2', 'streaming-general'), ('1', 'streaming-fused'), ('4', 'streaming-bg2'))] +
                [(d, e, 'tiny_res', 0, False) for d, e in [('1', 'streaming-general'), ('4', 'streaming-cf8'), ('4', 'streaming-bg2')]] +
                [(d, e, 'tiny_cat', 3, True) for d, e in [('4', 'streaming-general'), ('4', 'streaming'), ('1', 'streaming-fused')]] +
                [(d, e, 'tiny_tail', 1, False) for d, e in [('4', 'streaming'), ('4', 'streaming-bg8')]])


@pytest.mark.parametrize('depth,le_engine,name,seed,signed', _DEFER_CASES)
def test_deferred_stores_are_invisible(engine, monkeypatch, depth, le_engine, name, seed, signed):
    """Streaming engine, DFQ_LE_DEFER = depth: layers that are scaled one way only are stored every depth-th sweep and
    re-derived from the stored values and the remembered factors in between (dfq_le.hip).  Whatever the depth and however
    the sweeps are cut into enqueue calls, after every call the weights, the [O] vectors, the cumulative scales and the
    loop state are those of the reference loop stopped at that sweep -- bit for bit."""
    _select_le_engine(monkeypatch, le_engine)
    monkeypatch.setenv('DFQ_LE_DEFER', depth)
    model, graph, bottoms = synthetic.build(name, seed=seed)
    model.to(engine.device)
    lt.merge_batchnorm(model, graph, bottoms, TARG)
    rels = rel.create_relation(graph, bottoms, TARG)
    spec0 = graphspec.from_torch(graph, bottoms, TARG)
    plan = dfq.build_le_plan(graph, rels, TARG)
    assert plan.resident_tiles == 0
    if le_engine in ('streaming-general', 'streaming-persistent-3wg'):
        assert plan.defer_depth == int(depth) and plan.free_running_elements == 0 and plan.free_running_group == 1 and plan.lean_tiles == 0
        if depth == '1':
            assert plan.deferred_elements == 0
        else:
            assert 0 < plan.deferred_elements <= plan.rw_elements
            assert plan.sweep_bytes < 8 * plan.rw_elements + 4 * plan.ro_elements
    else:
        # the free-running segments (dfq_le_cf.hpp): layers whose every statistic is closed-form leave the sweep's launch
        group = {'streaming': 4, 'streaming-cf2': 2, 'streaming-cf8': 8, 'streaming-fused': 4, 'streaming-bg2': 2, 'streaming-bg4': 4,
                 'streaming-bg8': 8}[le_engine]
        if name in ('tiny_cat', 'tiny_tail'):   # (their chains run through dense layers scaled along both axes: nothing is free-running)
            assert plan.free_running_group == 1 and plan.free_running_elements == 0 and not plan.lean_background
        else:
            assert plan.free_running_group == group and plan.free_running_elements > 0 and plan.lean_tiles > 0
            assert plan.lean_background == le_engine.startswith('streaming-bg')
        assert plan.rw_elements + plan.free_running_elements <= plan.paired_elements
    plan.enqueue(0, restart=True, signed=signed)
    total = 0
    for n in (1, 1, 1, 2, 3, 5, 1, 1000):
        plan.enqueue(n, restart=False, signed=signed)
        total += n
        r = plan.query()
        plan.stage.writeback()
        spec = copy.deepcopy(spec0)
        n_o, S_o = orc.cross_layer_equalization(spec, orc.create_relation(spec), max_sweeps=total, signed=signed)
        assert r['sweeps'] == n_o, (total, r)
        osnap, esnap = _spec_snapshot(spec), snapshot(graph)
        for k in osnap:
            assert_bitexact(esnap[k], osnap[k], '{} after {} sweeps (depth {})'.format(k, total, depth))
        for a, b in zip(plan.scale_cum, S_o):
            assert_bitexact(npy(a), b, 'cumulative S after {} sweeps'.format(total))
    assert r['done']
    plan.close()


def test_block_info_accounts_for_every_element(engine, monkeypatch):
    """dfq_le_plan_block_info (tuning aid): the workgroups of a sweep launch cover every element the plan says a sweep
    reads and writes / only reads, exactly once."""
    monkeypatch.setenv('DFQ_LE_RESIDENT', '0')
    for name in ('tiny_mobile', 'tiny_res', 'tiny_cat'):
        model, graph, bottoms = synthetic.build(name, seed=0)
        model.to(engine.device)
        lt.merge_batchnorm(model, graph, bottoms, TARG)
        rels = rel.create_relation(graph, bottoms, TARG)
        plan = dfq.build_le_plan(graph, rels, TARG)
        rw = ro = 0
        for launch in range(plan.levels):
            for b in range(plan.level_info(launch)['workgroups']):
                info = plan.block_info(launch, b)
                assert 0 <= info['kind'] <= 5 and info['rows'] > 0 and info['cols'] > 0
                assert info['rw_elements'] + info['ro_elements'] == info['rows'] * info['cols']
                rw += info['rw_elements']
                ro += info['ro_elements']
        assert (rw, ro) == (plan.rw_elements, plan.ro_elements), name
        plan.close()


def test_resident_engine_in_chunks(engine):
    """enqueue(n) runs at most n sweeps per launch and carries the loop state: 3 + 3 + the rest == one run."""
    res = []
    for chunks in (None, (3, 3, 1000)):
        model, graph, bottoms = synthetic.build('tiny_mobile', seed=0)
        model.to(engine.device)
        lt.merge_batchnorm(model, graph, bottoms, TARG)
        rels = rel.create_relation(graph, bottoms, TARG)
        plan = dfq.build_le_plan(graph, rels, TARG)
        assert plan.resident_tiles > 0
        if chunks is None:
            r = plan.run()
        else:
            plan.enqueue(0, restart=True)
            for n in chunks:
                plan.enqueue(n, restart=False)
            r = plan.query()
            assert r.pop('done')
        plan.stage.writeback()
        res.append((r, snapshot(graph)))
        plan.close()
    assert res[0][0] == res[1][0]
    for k in res[0][1]:
        assert_bitexact(res[0][1][k], res[1][1][k], k)


# ---------------------------------------------------------------------------------------------------------------------
# opt-in lazy-scale equalisation: read-only sweeps from W0 and the cumulative scales, one final write
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name,seed,suffix,sweeps', [('tiny_mobile', 0, '', 7), ('tiny_res', 0, '', 3), ('tiny_cat', 0, '', 9),
                                                     ('tiny_mobile', 2, '_signed', 6), ('tiny_cat', 3, '_abs', 5)])
def test_lazy_scale_equalization_against_oracle(engine, name, seed, suffix, sweeps):
    """Same number of sweeps as the sequential loop -> every tensor and every cumulative scale within 1e-5 of the oracle's
    (the contract of BASELINE.json; the default engines are bit-exact, this formulation rounds its cumulative products
    differently).  Grouped / depthwise / signed / cat geometries of the fixtures."""
    gold = net_fixture(name, seed, suffix)
    model, graph, bottoms = _build(name, seed, gold, engine)
    lt.merge_batchnorm(model, graph, bottoms, TARG)
    spec = graphspec.from_torch(graph, bottoms, TARG)
    rels = rel.create_relation(graph, bottoms, TARG)
    signed = suffix == '_signed'
    _, S_ref = orc.cross_layer_equalization(spec, orc.create_relation(spec), max_sweeps=sweeps, converge_thres=-1.0,
                                            converge_count=10 ** 9, signed=signed)
    dfq.lazy_cross_layer_equalization(graph, rels, TARG, sweeps, signed=signed)
    worst = 0.0
    for i, k in enumerate(graph):
        n = spec.nodes[k]
        m = graph[k]
        if n.kind == 'targ':
            worst = max(worst, assert_close(npy(m.weight), n.weight, 'lazy w {}'.format(k)))
            if n.bias is not None and m.bias is not None:
                assert_close(npy(m.bias), n.bias, 'lazy b {}'.format(k))
        elif n.kind == 'bn' and n.fake_weight is not None:
            assert_close(npy(m.fake_weight), n.fake_weight, 'lazy gamma~ {}'.format(k))
            assert_close(npy(m.fake_bias), n.fake_bias, 'lazy beta~ {}'.format(k))
    for rr, s in zip(rels, S_ref):
        assert_close(npy(rr.get_scale_vec()), s, 'lazy S')
    assert worst <= 1e-5


def test_lazy_scale_batch_with_different_sweep_counts(engine):
    """Two networks in one plan, each with its own sweep count: each equals its own single-network run bit for bit."""
    outs = []
    for mode in ('batched', 'single'):
        nets = []
        for seed, sweeps in ((0, 3), (1, 6)):
            model, graph, bottoms = synthetic.build('tiny_mobile', seed=seed)
            model.to(engine.device)
            lt.merge_batchnorm(model, graph, bottoms, TARG)
            nets.append((graph, rel.create_relation(graph, bottoms, TARG), sweeps))
        if mode == 'batched':
            plan = dfq.LazyLEPlan([(g, r) for g, r, _ in nets], TARG)
            assert plan.paired_elements > 0 and plan.weight_elements > 0 and plan.levels >= 2
            plan.run([s for _, _, s in nets])
            from dfq_amd import _ffi
            _ffi.synchronize()
            plan.close()
        else:
            for g, r, s in nets:
                dfq.lazy_cross_layer_equalization(g, r, TARG, s)
        outs.append([snapshot(g) for g, _, _ in nets])
    for a, b in zip(outs[0], outs[1]):
        for k in a:
            assert_bitexact(a[k], b[k], 'lazy batched vs single: ' + k)


# ---------------------------------------------------------------------------------------------
# round 5: layers of SEVERAL tiles on the CPU emulation (statistics merged over row blocks: strict arrivals)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('tile_floats,spec,closed_form', [('8192', '2', False), ('2048', '0', False), ('1024', '4', False),
                                                          ('2048', '0', True)])
def test_column_statistics_merged_over_row_blocks(engine, monkeypatch, tile_floats, spec, closed_form):
    """A layer cut into several row blocks merges its per-input-channel (min, max) over them every sweep (atomicMax into shared
    tagged words, a strict arrival on the layer's counter).  Until round 5 only the full-size networks had such layers, i.e. only the
    GPU ran that code.  `tiny_tail` has one at the default tile size (2 row blocks of its 200 x 48 layer); DFQ_RES_TILE_FLOATS (a test
    knob) shrinks the tiles so that it is cut into 13 / 7 row blocks and the classifier into 3 / 2.  Whatever the cut, the speculation
    depth and the chunking of the loop into launches: weights, [O] vectors, cumulative scales and the sweep count are the oracle's,
    bit for bit.  (The same merge through per-tile slots -- plain stores, every tile reduces its slice, tagged polls -- was built,
    passed this test and was measured SLOWER on the MI355X: 0.73 vs 0.61 ms for MobileNetV2)"""
    _select_le_engine(monkeypatch, 'resident-cf' if closed_form else 'resident')   # (closed form: one tile per channel publishes)
    monkeypatch.setenv('DFQ_RES_TILE_FLOATS', tile_floats)
    monkeypatch.setenv('DFQ_RES_SPEC', spec)

    def fresh():
        model, graph, bottoms = synthetic.build('tiny_tail', seed=0)
        model.to(engine.device)
        lt.merge_batchnorm(model, graph, bottoms, TARG)
        return model, graph, bottoms, rel.create_relation(graph, bottoms, TARG)

    def check(graph, plan, sp, S_o):
        snap = snapshot(graph)
        for i, k in enumerate(graph):
            n = sp.nodes[k]
            if n.kind == 'targ':
                assert_bitexact(snap['L{}.w'.format(i)], n.weight, 'w {}'.format(k))
                if n.bias is not None and 'L{}.b'.format(i) in snap:
                    assert_bitexact(snap['L{}.b'.format(i)], n.bias, 'b {}'.format(k))
            elif n.kind == 'bn' and n.fake_weight is not None:
                assert_bitexact(snap['L{}.fw'.format(i)], n.fake_weight, 'fw {}'.format(k))
                assert_bitexact(snap['L{}.fb'.format(i)], n.fake_bias, 'fb {}'.format(k))
        for a, b in zip(plan.scale_cum, S_o):
            assert_bitexact(npy(a), b, 'cumulative S')

    model, graph, bottoms, rels = fresh()
    spec0 = graphspec.from_torch(graph, bottoms, TARG)
    orels = orc.create_relation(spec0)
    assert len(rels) == 4
    plan = dfq.build_le_plan(graph, rels, TARG)
    assert plan.resident_tiles >= {'8192': 6, '2048': 11, '1024': 18}[tile_floats], (plan.resident_tiles, plan.resident_reason)
    # (1) the data-dependent loop in one launch
    res = plan.run()
    plan.stage.writeback()
    sp = spec0.clone()
    n_o, S_o = orc.cross_layer_equalization(sp, orels)
    assert res['sweeps'] == n_o == 34
    check(graph, plan, sp, S_o)
    plan.close()
    # (2) cut into launches of 1, 3, 5, ... sweeps
    model, graph, bottoms, rels = fresh()
    plan = dfq.build_le_plan(graph, rels, TARG)
    plan.enqueue(0, restart=True)
    total, step = 0, 1
    while total < n_o:
        plan.enqueue(step, restart=False)
        total = min(n_o, total + step)
        q = plan.query()
        assert q['sweeps'] == total
        step += 2
    plan.stage.writeback()
    check(graph, plan, sp, S_o)
    plan.close()


@pytest.mark.gpu
def test_default_stream_is_an_ordinary_stream(engine):
    """The drop-in entry points run on torch's current stream, for most callers the NULL stream.  It gets the same protocol
    (tagged slots, guarded against another stream's in-launch waits) and the same result as a side stream."""
    import torch
    gold = net_fixture('tiny_mobile', 0, '')
    results = []
    for side in (False, True):
        model, graph, bottoms = _build('tiny_mobile', 0, gold, engine)
        lt.merge_batchnorm(model, graph, bottoms, TARG)
        load_stage(graph, gold, 'abs')
        stream = torch.cuda.Stream() if side else torch.cuda.default_stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            plan, _ = dfq.build_bc_plan(graph, bottoms, TARG)
            plan.run(check=True)
            assert plan.tagged and plan.last_run_tagged, 'side' if side else 'default'
            plan.close()
        torch.cuda.synchronize()
        results.append(snapshot(graph))
    for k in results[0]:
        assert_bitexact(results[0][k], results[1][k], k)


@pytest.mark.parametrize('setting', ['two-launches', 'ahead-0', 'ahead-100'])
def test_one_launch_correction_is_invisible(engine, monkeypatch, setting):
    """Round 5: a tagged correction can be ONE launch (the default for a batch) -- the per-tensor min/max blocks are workgroups of the chain launch, woven in
    `DFQ_BC_MM_AHEAD` (default 2) chain positions in front of the steps that need them; a step waits for its layer's blocks
    through a per-layer arrival counter.  Min and max do not care who forms them: results BIT-IDENTICAL to the pair of launches
    (DFQ_BC_ONE_LAUNCH=0), for the extreme look-aheads (0: right in front of the position; 100: all of them first), signed and unsigned,
    and a second run of the plan (the other parity of the slots and counters) repeats the first."""
    for k in ('DFQ_BC_TAGGED', 'DFQ_BC_MERGED', 'DFQ_BC_FOLD', 'DFQ_BC_ONE_LAUNCH', 'DFQ_BC_MM_AHEAD'):
        monkeypatch.delenv(k, raising=False)
    for name, seed, suffix in [('tiny_mobile', 0, ''), ('tiny_mobile', 2, '_signed'), ('tiny_cat', 0, ''), ('tiny_res', 0, '')]:
        gold = net_fixture(name, seed, suffix)
        signed = suffix == '_signed'
        results = {}
        for variant in ('default', setting):
            monkeypatch.setenv('DFQ_BC_ONE_LAUNCH', '1')          # (the default for batches only; forced for these single networks)
            monkeypatch.delenv('DFQ_BC_MM_AHEAD', raising=False)
            if variant == 'two-launches':
                monkeypatch.setenv('DFQ_BC_ONE_LAUNCH', '0')
            elif variant.startswith('ahead-'):
                monkeypatch.setenv('DFQ_BC_MM_AHEAD', variant.split('-')[1])
            model, graph, bottoms = _build(name, seed, gold, engine)
            lt.merge_batchnorm(model, graph, bottoms, TARG)
            load_stage(graph, gold, 'abs')
            plan, _ = dfq.build_bc_plan(graph, bottoms, TARG)
            assert plan.one_launch == (variant != 'two-launches'), (name, variant)
            plan.run(signed=signed, check=True)
            first = snapshot(graph)
            compare_stage(first, gold, 'bc', what='{} {}'.format(name, variant))
            load_stage(graph, gold, 'abs')
            plan.run(signed=signed, check=True)
            for k, v in snapshot(graph).items():
                assert_bitexact(v, first[k], '{} second run {} ({})'.format(name, k, variant))
            plan.close()
            results[variant] = first
        for k, v in results['default'].items():
            assert_bitexact(v, results[setting][k], '{} {} ({} vs default)'.format(name, k, setting))


@pytest.mark.gpu
def test_one_launch_correction_under_stress(monkeypatch):
    """everything handed over behind the arrival counters of the one-launch correction is written with
    device-scope atomics / sc1 stores and read with device-scope loads -- no release / acquire pair (BcFusedMm, dfq_bc.hip) -- an
    invariant only the hardware can check (the CPU emulation's atomics are sequentially consistent).  A full-size batch, the
    look-ahead at 0 (the steps find their layer's min/max blocks directly in front of them: every hand-over is as late as it can be),
    forty runs: every run must leave the bits of the two-launch correction."""
    from dfq_amd import synthetic
    dev = torch.device('cuda', 0)

    def batch():
        nets = []
        for seed in range(4):
            model, graph, bottoms = synthetic.build('mobilenet_v2', seed=seed)
            synthetic.relu6_to_relu(model)
            model.to(dev)
            lt.merge_batchnorm(model, graph, bottoms, TARG)
            nets.append((model, graph, bottoms))
        return nets

    def run(nets, reps):
        bc = dfq.build_bc_plan_batch([(g, b) for (_, g, b) in nets], TARG)
        saved = [[(m.bias.clone() if getattr(m, 'bias', None) is not None else None, getattr(m, 'fake_bias', None).clone() if hasattr(m, 'fake_bias') else None)
                  for m in g.values() if not isinstance(m, str)] for (_, g, _) in nets]
        outs = []
        for _ in range(reps):
            with torch.no_grad():
                for (_, g, _), sv in zip(nets, saved):
                    for m, (b, fb) in zip([m for m in g.values() if not isinstance(m, str)], sv):
                        if b is not None:
                            m.bias.copy_(b)
                        if fb is not None:
                            m.fake_bias.copy_(fb)
            bc.run(check=True, recover=False)
            outs.append([snapshot(g) for (_, g, _) in nets])
        one = bc.one_launch
        bc.close()
        return outs, one

    monkeypatch.setenv('DFQ_BC_ONE_LAUNCH', '0')
    (want,), one = run(batch(), 1)
    assert not one
    monkeypatch.delenv('DFQ_BC_ONE_LAUNCH')
    monkeypatch.setenv('DFQ_BC_MM_AHEAD', '0')
    outs, one = run(batch(), 40)
    assert one
    for rep, got in enumerate(outs):
        for a, b in zip(got, want):
            for k in b:
                assert_bitexact(a[k], b[k], 'run {}: {}'.format(rep, k))
