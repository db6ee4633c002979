"""A property the next step of the single-network engine rests on ("statistics that need no exchange"):
a layer that a sweep scales UNIFORMLY along the axis its statistic reduces over has that statistic in closed form --
multiplication by s > 0 and float32 rounding are monotonic, so max_i fl(w_i * s) == fl(max_i w_i * s) bit for bit, sweep after
sweep.  For an inverted-residual block (1x1 expand -> 3x3 depthwise -> 1x1 project, relations (a, b) and (b, c) as the
reference pairs them, dfq.py:29-75) that is every statistic the two relations consume: the scale factors of ALL sweeps follow
from six scalars per channel taken once, with no pass over the weights.  Checked here against the oracle's element-wise
equalisation (oracle/dfq_oracle.py layer_equalization, itself pinned to the reference), signed and unsigned, with a dead
channel and with denormal-range values."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dfq_oracle as orc

F32 = np.float32


def _block(seed, ch=48, cin=16, cout=24):
    g = np.random.default_rng(seed)
    a = (g.standard_normal((ch, cin, 1, 1)) * g.uniform(0.01, 3.0, (ch, 1, 1, 1))).astype(F32)      # expand: rows = channels
    b = (g.standard_normal((ch, 1, 3, 3)) * g.uniform(0.05, 2.0, (ch, 1, 1, 1))).astype(F32)        # depthwise
    c = (g.standard_normal((cout, ch, 1, 1)) * g.uniform(0.02, 1.5, (1, ch, 1, 1))).astype(F32)     # project: columns = channels
    a[5] = 0.0                                   # a dead channel (range 0: the clamp branch of dfq.py:58-59)
    b[7] *= F32(1e-38)                           # products in the denormal range
    return a, b, c


def _stats(x2d, signed):
    """(lo, hi) per row of a [channels, n] view: what a range is made of"""
    if signed:
        return np.zeros(x2d.shape[0], F32), np.abs(x2d).max(-1).astype(F32)
    return x2d.min(-1).astype(F32), x2d.max(-1).astype(F32)


def _range(lo, hi, signed):
    return hi.copy() if signed else (hi - lo).astype(F32)


@pytest.mark.parametrize('signed', [False, True])
@pytest.mark.parametrize('seed', [0, 1, 2])
def test_block_scales_follow_from_six_scalars_per_channel(seed, signed):
    a, b, c = _block(seed)
    ch = a.shape[0]
    # ---- the scalars, taken once from the untouched weights ----
    lo_a, hi_a = _stats(a.reshape(ch, -1), signed)
    lo_b, hi_b = _stats(b.reshape(ch, -1), signed)
    lo_c, hi_c = _stats(np.transpose(c.reshape(c.shape[0], ch), (1, 0)), signed)
    for sweep in range(40):
        # ---- the oracle: element-wise, as the reference does it ----
        s1 = orc.layer_equalization(a, b, None, signed=signed)
        s2 = orc.layer_equalization(b, c, None, signed=signed)
        # ---- the recurrence: no weight is read ----
        with np.errstate(all='ignore'):
            q1, inv1 = orc.le_solve(_range(lo_a, hi_a, signed), _range(lo_b, hi_b, signed))
            lo_a, hi_a = (lo_a * q1).astype(F32), (hi_a * q1).astype(F32)
            lo_b, hi_b = (lo_b * inv1).astype(F32), (hi_b * inv1).astype(F32)
            q2, inv2 = orc.le_solve(_range(lo_b, hi_b, signed), _range(lo_c, hi_c, signed))
            lo_b, hi_b = (lo_b * q2).astype(F32), (hi_b * q2).astype(F32)
            lo_c, hi_c = (lo_c * inv2).astype(F32), (hi_c * inv2).astype(F32)
        assert np.array_equal(s1.view(np.int32), q1.view(np.int32)), (sweep, 'relation (a, b)')
        assert np.array_equal(s2.view(np.int32), q2.view(np.int32)), (sweep, 'relation (b, c)')
    # ... and the scalars ARE the statistics of the element-wise result
    for (lo, hi), x in (((lo_a, hi_a), a.reshape(ch, -1)), ((lo_b, hi_b), b.reshape(ch, -1)),
                        ((lo_c, hi_c), np.transpose(c.reshape(c.shape[0], ch), (1, 0)))):
        tl, th = _stats(x, signed)
        # (value equality: the maximum over a column of zeros of both signs is a zero of either sign -- its range is 0 all the same)
        assert np.array_equal(tl, lo) and np.array_equal(th, hi)


def test_a_layer_scaled_along_both_axes_has_no_closed_form():
    """The counter-example that keeps the exchanges of MobileNetV2's tail: a 1x1 layer that is the SECOND layer of one relation
    (columns scaled by 1/s[k]) and the FIRST of the next (rows scaled) -- its row maxima after the column scaling are not the old
    maxima times anything."""
    g = np.random.default_rng(3)
    w = g.standard_normal((8, 12)).astype(F32)
    inv = g.uniform(0.2, 5.0, 12).astype(F32)
    scaled = (w * inv[None, :]).astype(F32)
    ratios = scaled.max(-1) / w.max(-1)
    assert ratios.max() / ratios.min() > 1.5


# ---- ... and against the unmodified reference (oracle/_ref: dfq.py byte-compiled from /root/reference by oracle/build_ref.py) ----
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, 'oracle', '_ref')


@pytest.mark.skipif(not os.path.isfile(os.path.join(REFDIR, 'dfq.pyc')),
                    reason='oracle/_ref is not built (oracle/build_ref.py needs /root/reference)')
@pytest.mark.parametrize('signed', [False, True])
def test_block_recurrence_against_the_reference_itself(signed):
    """The same recurrence against the REFERENCE's `_layer_equalization` (dfq.py:29-75, torch on the CPU): the scale vectors of
    both relations over 30 sweeps.  torch's CPU sqrt is not correctly rounded, so scale factors may differ from numpy's in the
    last place -- the recurrence is therefore fed the reference's OWN solve, one channel at a time as the reference
    does it: what is checked is that ranges taken from six scalars equal ranges taken from the weights, sweep after sweep."""
    import torch
    before = set(sys.modules)
    sys.path.insert(0, REFDIR)
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    try:
        import dfq as ref_dfq
        assert os.path.dirname(os.path.abspath(ref_dfq.__file__)) == REFDIR
        a, b, c = (torch.from_numpy(x.copy()) for x in _block(4))
        ch = a.shape[0]

        def stats(x2d):
            if signed:
                return torch.zeros(x2d.shape[0]), x2d.abs().max(-1)[0]
            return x2d.min(-1)[0], x2d.max(-1)[0]

        def rng(lo, hi):
            return hi.clone() if signed else hi - lo

        def solve(r1, r2):                      # dfq.py:56-59, element by element like the reference's loop
            s = torch.empty(ch)
            inv = torch.empty(ch)
            for k in range(ch):
                v = (1 / (r1[k] + 0)) * torch.sqrt(r1[k] * r2[k] + 0)
                v = max(1e-8, min(1e8, v))
                s[k] = v
                inv[k] = 1 / v
            return s, inv

        lo_a, hi_a = stats(a.reshape(ch, -1))
        lo_b, hi_b = stats(b.reshape(ch, -1))
        lo_c, hi_c = stats(c.reshape(c.shape[0], ch).t())
        with torch.no_grad():
            for sweep in range(30):
                _, _, _, s1 = ref_dfq._layer_equalization(a, b, None, signed=signed)
                _, _, _, s2 = ref_dfq._layer_equalization(b, c, None, signed=signed)
                q1, inv1 = solve(rng(lo_a, hi_a), rng(lo_b, hi_b))
                lo_a, hi_a, lo_b, hi_b = lo_a * q1, hi_a * q1, lo_b * inv1, hi_b * inv1
                q2, inv2 = solve(rng(lo_b, hi_b), rng(lo_c, hi_c))
                lo_b, hi_b, lo_c, hi_c = lo_b * q2, hi_b * q2, lo_c * inv2, hi_c * inv2
                assert torch.equal(s1, q1), (sweep, 'relation (a, b)')
                assert torch.equal(s2, q2), (sweep, 'relation (b, c)')
    finally:
        sys.dont_write_bytecode = old
        sys.path.remove(REFDIR)
        for name in set(sys.modules) - before:
            if name == 'dfq' or name == 'utils' or name.startswith('utils.'):
                del sys.modules[name]
