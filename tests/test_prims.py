"""Stand-alone primitives of the C ABI against the numpy oracle.

Integer / min-max / IEEE-elementwise work is bit-exact; the float64-accumulated reductions are compared
after their single rounding to float32 (bit-exact unless the float64 sums differ in the last place, hence
1-ulp tolerance).  Runs on the CPU emulation of the kernels and, marked gpu, on the MI355X."""
import numpy as np
import pytest
import torch

from dfq_amd import prims
from oracle import dfq_oracle as orc
from tests.common import assert_bitexact, npy

PAIRS = [
    ((24, 16, 1, 1), (40, 24, 1, 1)),        # 1x1 -> 1x1
    ((32, 1, 3, 3), (16, 32, 1, 1)),         # depthwise first
    ((16, 8, 1, 1), (16, 1, 5, 5)),          # depthwise second (grouped pairing)
    ((12, 7, 3, 3), (10, 12)),               # conv -> linear
    ((70, 130), (9, 70)),                    # linear -> linear, rows longer than a wave
]


def _rand(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * torch.rand(shape[0], *([1] * (len(shape) - 1)), generator=g).add(0.05)


@pytest.mark.parametrize('s1,s2', PAIRS)
@pytest.mark.parametrize('signed', [False, True])
def test_ranges_solve_apply_pair(engine, s1, s2, signed):
    w1, w2 = _rand(s1, 1), _rand(s2, 2)
    w1[3] = 0.0                                            # dead channel: r1 == 0 -> S = 1e8 (NaN through the clamp)
    b1, bnw, bnb = torch.randn(s1[0]), torch.rand(s1[0]) + 0.5, torch.randn(s1[0])
    a1, w2v, G, gi, go = orc._pair_views(w1.numpy().copy(), w2.numpy().copy())
    cols = np.transpose(w2v, (0, 2, 1, 3)).reshape(s1[0], -1)
    r1o, r2o = orc.channel_ranges(a1, signed), orc.channel_ranges(cols, signed)
    r1 = prims.row_range(engine.to(w1), signed)
    r2 = prims.col_range(engine.to(w2), s1[0], signed)
    assert_bitexact(npy(r1), r1o, 'row ranges')
    assert_bitexact(npy(r2), r2o, 'column ranges')
    so, invo = orc.le_solve(r1o, r2o)
    s, inv = prims.le_solve(r1, r2)
    assert_bitexact(npy(s), so, 'S')
    assert_bitexact(npy(inv), invo, '1/S')
    assert float(npy(s)[3]) == np.float32(1e8)

    # apply == oracle's in-place pair update; pair == the same through one entry point
    ow1, ow2, ob1, obw, obb = [t.numpy().copy() for t in (w1, w2, b1, bnw, bnb)]
    oS = orc.layer_equalization(ow1, ow2, ob1, obw, obb, signed=signed)
    for mode in ('apply', 'pair'):
        e = [engine.to(t.clone()) for t in (w1, w2, b1, bnw, bnb)]
        if mode == 'apply':
            prims.le_apply(e[0], e[1], e[2], e[3], e[4], s, inv)
        else:
            S = prims.le_pair(e[0], e[1], e[2], e[3], e[4], signed=signed)
            assert_bitexact(npy(S), oS, 'S of le_pair')
        for got, want, what in zip(e, (ow1, ow2, ob1, obw, obb), ('W1', 'W2', 'b1', 'bn_weight', 'bn_bias')):
            assert_bitexact(npy(got), want, '{} after le_{}'.format(what, mode))


def test_le_pair_on_cpu_tensors_stages_and_writes_back(engine):
    w1, w2 = _rand((8, 4, 3, 3), 5), _rand((6, 8, 1, 1), 6)
    ow1, ow2 = w1.numpy().copy(), w2.numpy().copy()
    oS = orc.layer_equalization(ow1, ow2, None)
    S = prims.le_pair(w1, w2, None)                        # CPU tensors: staged, written back
    assert S.device.type == 'cpu'
    assert_bitexact(S.numpy(), oS, 'S')
    assert_bitexact(w1.numpy(), ow1, 'W1')
    assert_bitexact(w2.numpy(), ow2, 'W2')


@pytest.mark.parametrize('n', [1, 255, 4096, 70001])
def test_absdiff_mean(engine, n):
    g = torch.Generator().manual_seed(n)
    a = torch.randn(n, generator=g)
    b = a + torch.randn(n, generator=g) * 1e-3
    got = prims.absdiff_mean(engine.to(a), engine.to(b))
    want = orc.layer_absdiff_mean(a.numpy(), b.numpy())
    assert abs(got - want) <= abs(want) * 2e-7
    assert prims.absdiff_mean(engine.to(a), engine.to(a.clone())) == 0.0


@pytest.mark.parametrize('shape,bits,sym', [((5, 3, 3, 3), 8, False), ((7, 130), 8, True), ((3, 1, 1, 1), 4, False),
                                            ((64, 9), 16, True)])
def test_fake_quant_rows_matches_per_row_oracle(engine, shape, bits, sym):
    x = _rand(shape, 11)
    y, codes, mm = prims.fake_quant_rows(engine.to(x), bits, symmetric=sym, return_codes=True)
    xn = x.numpy().reshape(shape[0], -1)
    for r in range(shape[0]):
        mn, mx = float(xn[r].min()), float(xn[r].max())
        assert npy(mm)[r, 0] == np.float32(mn) and npy(mm)[r, 1] == np.float32(mx)
        want, wcodes = orc.uniform_quantize(xn[r], bits, mn, mx, symmetric=sym, return_codes=True)
        assert_bitexact(npy(y).reshape(shape[0], -1)[r], want, 'row {}'.format(r))
        assert np.array_equal(npy(codes).reshape(shape[0], -1)[r], wcodes.astype(np.float32))
    # explicit ranges: every row with the tensor-wide range == the per-tensor quantiser
    lo = torch.full((shape[0],), float(xn.min()))
    hi = torch.full((shape[0],), float(xn.max()))
    y2 = prims.fake_quant_rows(engine.to(x), bits, engine.to(lo), engine.to(hi), symmetric=sym)
    want = orc.uniform_quantize(x.numpy(), bits, float(xn.min()), float(xn.max()), symmetric=sym)
    assert_bitexact(npy(y2), want, 'rows with a shared range')


@pytest.mark.parametrize('O,Ig,groups', [(12, 40, 1), (32, 1, 32), (16, 6, 4), (5, 300, 1)])
def test_grouped_matvec(engine, O, Ig, groups):
    g = torch.Generator().manual_seed(O * 7 + Ig)
    eps = torch.randn(O, Ig, generator=g) * 1e-3
    ex = torch.randn(groups * Ig, generator=g)
    got = npy(prims.grouped_matvec(engine.to(eps), engine.to(ex), groups))
    step = O // groups
    want = np.concatenate([orc._matvec_f32(eps.numpy()[k * step:(k + 1) * step], ex.numpy()[k * Ig:(k + 1) * Ig])
                           for k in range(groups)])
    np.testing.assert_allclose(got, want, rtol=2e-7, atol=0)


def test_zeroq_per_channel_quantiser(engine):
    """ZeroQ's per-output-channel asymmetric quantiser: bit-exact against the reference's outputs (fixture written by
    oracle/make_golden_minmax.py from the unmodified AsymmetricQuantFunction) and against the oracle incl. codes."""
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'kat_zeroq_rows.npz'))
    i = 0
    while 'x{}'.format(i) in gold.files:
        x, bits = gold['x{}'.format(i)], int(gold['bits{}'.format(i)])
        y, codes = prims.zeroq_quant_rows(engine.to(torch.from_numpy(x.copy())), bits, return_codes=True)
        assert_bitexact(npy(y), gold['y{}'.format(i)], 'zeroq case {} vs reference'.format(i))
        oy, oq = orc.zeroq_quant_rows(x, bits, return_codes=True)
        assert_bitexact(oy, gold['y{}'.format(i)], 'oracle vs reference')
        assert np.array_equal(npy(codes), oq)
        assert npy(codes).min() >= -2 ** (bits - 1) and npy(codes).max() <= 2 ** (bits - 1) - 1
        i += 1
    assert i >= 5
