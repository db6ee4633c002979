"""CPU oracle for the DFQ calibration hot path (numpy, float32 arithmetic).

TEST INFRASTRUCTURE ONLY.  Nothing under ``dfq_amd/`` may import this module.  It is
imported by ``tests/``, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of
``bench.py`` -- always as the checker / the timed CPU baseline, never as the product path.

It restates, on plain numpy arrays, the algorithm of the reference (jakc4103/DFQ):

  * ``uniform_quantize``        <- utils/quantize.py:23-76   (UniformQuantize.forward)
  * ``quantize_error``          <- dfq.py:8-25               (_quantize_error)
  * ``layer_equalization``      <- dfq.py:28-75              (_layer_equalization)
  * ``cross_layer_equalization``<- dfq.py:78-117
  * ``bias_absorption``         <- dfq.py:121-164
  * ``clip_weight``             <- dfq.py:167-170
  * ``find_prev_bn``            <- utils/layer_transform.py:299-344
  * ``bias_correction``         <- dfq.py:173-293
  * ``merge_batchnorm``         <- utils/layer_transform.py:231-276
  * ``quantize_targ_layer``     <- utils/layer_transform.py:279-296
  * ``create_relation``         <- utils/relation.py:30-94
  * ``quant_measure_update``    <- utils/quantize.py:102-119 (QuantMeasure.forward)
  * ``merge_scale_to_weight``   <- utils/quantize.py:145-174, :269-289

Parity pinning: ``oracle/make_golden.py`` imports the *unmodified* reference from
``/root/reference`` (possible only in the build container), runs both on the same seeded inputs
and asserts agreement before it writes the fixtures in ``tests/golden/``.  Agreement is bit-exact
for the fake-quant round trip and for every LE quantity except where this torch CPU build's
``torch.sqrt`` is not correctly rounded (about 0.3 % of channels differ by 1 ulp in S, see
DESIGN.md); those are held to 1e-5.  The oracle itself uses IEEE-correct float32 operations only
(numpy elementwise ops), which is also what the HIP kernels use, so HIP-vs-oracle comparisons on
LE / fake-quant are exact.

Graph model used here (pure python, no torch): ``GraphSpec`` -- see ``oracle/graphspec.py``.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------------------------
# a4  UniformQuantize.forward  (utils/quantize.py:23-76)
# --------------------------------------------------------------------------------------------
def quant_params(num_bits, min_value, max_value, symmetric):
    """Scalar part of the recipe when min/max arrive as Python floats (quantize.py:49-66).

    Everything here is Python double arithmetic, exactly as in the reference; the float32
    casts happen where torch casts a Python scalar operand to the tensor dtype.
    Returns (qmin, qmax, min_value(double), scale(double)).
    """
    if symmetric:
        qmin = -2.0 ** (num_bits - 1)
        qmax = 2 ** (num_bits - 1) - 1
        max_value = abs(max_value)
        min_value = abs(min_value)
        if max_value < min_value:
            max_value = min_value
        scale = max_value / qmax
        min_value = 0.0
    else:
        qmin = 0.0
        qmax = 2.0 ** num_bits - 1.0
        scale = (max_value - min_value) / (qmax - qmin)
    scale = max(scale, 1e-8)
    return float(qmin), float(qmax), float(min_value), float(scale)


def fake_quant_f32(x, qmin, qmax, neg_min_f32, scale_f32, min_f32, return_codes=False):
    """The five elementwise float32 passes of quantize.py:70-74 given ready-made float32 scalars.

    out = rint(clip((x + (-min)) / scale, qmin, qmax)) * scale + min   -- each op rounded to f32.
    """
    x = np.asarray(x, dtype=F32)
    q = (x + F32(neg_min_f32)).astype(F32)
    q = (q / F32(scale_f32)).astype(F32)
    q = np.clip(q, F32(qmin), F32(qmax))
    q = np.rint(q).astype(F32)                      # round half to even == torch.round_
    out = (q * F32(scale_f32)).astype(F32)
    out = (out + F32(min_f32)).astype(F32)
    if return_codes:
        return out, q
    return out


def uniform_quantize(x, num_bits=8, min_value=None, max_value=None, symmetric=False,
                     num_chunks=None, return_codes=False):
    """UniformQuantize.forward (quantize.py:23-76) on a numpy array.

    Two recipes exist in the reference and both are restated:
      * min/max given as Python floats  -> scale computed in double, cast to f32 per op;
      * min/max None                    -> per-chunk-row min/max mean as 0-dim float32 tensors and
                                           the whole scale computation stays in float32
                                           (quantize.py:24-35).
    """
    x = np.asarray(x, dtype=F32)
    tensor_path = (min_value is None) or (max_value is None)
    if tensor_path:
        B = x.shape[0]
        nc = B if num_chunks is None else num_chunks
        y = x.reshape(B // nc, -1)
        if min_value is None:
            min_value = _mean_f32(y.min(-1))
        if max_value is None:
            max_value = _mean_f32(y.max(-1))
        # 0-dim float32 tensor arithmetic (python scalars are cast to f32 by torch)
        mn = F32(min_value)
        mx = F32(max_value)
        if symmetric:
            qmin = -2.0 ** (num_bits - 1)
            qmax = 2 ** (num_bits - 1) - 1
            mx = F32(abs(mx))
            mn = F32(abs(mn))
            if mx < mn:
                mx = mn
            scale = F32(mx / F32(qmax))
            mn_used = F32(0.0)
        else:
            qmin = 0.0
            qmax = 2.0 ** num_bits - 1.0
            scale = F32(F32(mx - mn) / F32(qmax - qmin))
            mn_used = mn
        # max(scale, 1e-8): tensor > python float compares in f32 after casting 1e-8 ... the
        # reference keeps the tensor when it is the larger one, else the python float.
        if scale < F32(1e-8):                            # python max(scale, 1e-8): NaN is kept
            scale = F32(1e-8)
        neg_min = F32(-mn_used)
        return fake_quant_f32(x, qmin, qmax, neg_min, scale, mn_used, return_codes)

    qmin, qmax, mn_d, scale_d = quant_params(num_bits, float(min_value), float(max_value), symmetric)
    return fake_quant_f32(x, qmin, qmax, F32(-mn_d), F32(scale_d), F32(mn_d), return_codes)


def _mean_f32(v):
    """torch .mean(-1) of a short float32 vector; for one element it is the element itself."""
    v = np.asarray(v, dtype=F32)
    if v.size == 1:
        return F32(v.reshape(-1)[0])
    return F32(np.float64(v.astype(np.float64).sum()) / v.size)


# --------------------------------------------------------------------------------------------
# a3  _quantize_error (dfq.py:8-25)
# --------------------------------------------------------------------------------------------
def quantize_error(param, num_bits=8, reduction='sum', signed=False):
    param = np.asarray(param, dtype=F32)
    q = uniform_quantize(param, num_bits, float(param.min()), float(param.max()), signed)
    eps = (q - param).astype(F32)
    if reduction == 'sum':
        return F32(np.abs(eps).astype(np.float64).sum())
    if reduction == 'mean':
        return F32(eps.astype(np.float64).mean())
    if reduction == 'channel':
        return F32(np.abs(eps.reshape(eps.shape[0], -1).astype(np.float64).sum(-1)).sum())
    if reduction == 'spatial':
        return F32(np.abs(eps.reshape(eps.shape[0], eps.shape[1], -1).astype(np.float64).sum(-1)).sum())
    return eps


def quant_error_rowsum(weight, signed=False, num_bits=8):
    """eps.view(O, I, -1).sum(-1) of dfq.py:218-219, float32 sequential sum over kH*kW."""
    w = np.asarray(weight, dtype=F32)
    eps = quantize_error(w, num_bits, None, signed)
    e3 = eps.reshape(w.shape[0], w.shape[1], -1)
    acc = np.zeros(e3.shape[:2], dtype=F32)
    for k in range(e3.shape[2]):                       # fixed left-to-right f32 order
        acc = (acc + e3[:, :, k]).astype(F32)
    return acc


# --------------------------------------------------------------------------------------------
# a1  _layer_equalization (dfq.py:28-75), vectorised over channels (channels are independent)
# --------------------------------------------------------------------------------------------
def _pair_views(w1, w2):
    """Views pairing channel c of W1 with its column in W2 (dfq.py:29-46).

    Returns a1 [O1, row_len] (view of w1) and a2 [O1, go*khkw] (a COPY gathered from w2) plus
    the index bookkeeping needed to scatter a2 back.
    """
    O1 = w1.shape[0]
    I2g = w2.shape[1]
    num_group = 1
    if O1 != I2g:
        num_group = O1 // I2g
    gi = O1 // num_group
    go = w2.shape[0] // num_group
    a1 = w1.reshape(O1, -1)
    w2v = w2.reshape(num_group, go, I2g, -1)            # [G, go, I2g, khkw]
    return a1, w2v, num_group, gi, go


def le_solve(r1, r2, s_range=(1e-8, 1e8), eps=0):
    """s = (1/(r1+eps)) * sqrt(r1*r2+eps) in float32, then the Python clamp of dfq.py:58-59.

    Python ``max(lo, min(hi, s))`` semantics on a 0-dim tensor: ``min(hi, s)`` returns ``s`` only
    if ``s < hi`` is True, else ``hi``;  ``max(lo, t)`` returns ``t`` only if ``t > lo``.
    NaN compares False both times -> NaN becomes ``hi`` (dead channel r1 == 0 -> 1e8).
    Returns (s_f32, inv_f32) where inv is the float32 value the reference multiplies W2 by:
    ``1/s`` is a float32 reciprocal for a tensor ``s`` and a double reciprocal cast to float32
    when the clamp replaced ``s`` by a Python float.
    """
    r1 = np.asarray(r1, dtype=F32)
    r2 = np.asarray(r2, dtype=F32)
    e = F32(eps)
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        s = ((F32(1.0) / (r1 + e).astype(F32)).astype(F32)
             * np.sqrt(((r1 * r2).astype(F32) + e).astype(F32)).astype(F32)).astype(F32)
        lo, hi = float(s_range[0]), float(s_range[1])
        keep_hi = s < F32(hi)                            # False for NaN
        t = np.where(keep_hi, s, F32(hi)).astype(F32)
        # comparison `t > lo`: tensor vs python float -> float32 compare for tensor t;
        # for the python float hi it is a double compare, which hi > lo satisfies.
        keep_lo = np.where(keep_hi, t > F32(lo), hi > lo)
        s_out = np.where(keep_lo, t, F32(lo)).astype(F32)
        inv_tensor = (F32(1.0) / s_out).astype(F32)
        inv_hi = F32(1.0 / hi)
        inv_lo = F32(1.0 / lo)
        inv = np.where(keep_lo, np.where(keep_hi, inv_tensor, inv_hi), inv_lo).astype(F32)
    return s_out, inv


def channel_ranges(a, signed):
    if signed:
        return np.abs(a).max(-1).astype(F32)
    return (a.max(-1) - a.min(-1)).astype(F32)


def layer_equalization(w1, w2, b1, bn_weight=None, bn_bias=None, s_range=(1e-8, 1e8),
                       signed=False, eps=0):
    """In-place on the numpy arrays; returns S (float32 [O1])."""
    a1, w2v, G, gi, go = _pair_views(w1, w2)
    O1 = a1.shape[0]
    # column of channel c=(g,ii): w2v[g, :, ii, :]
    cols = np.transpose(w2v, (0, 2, 1, 3)).reshape(O1, -1)   # copy [O1, go*khkw]
    r1 = channel_ranges(a1, signed)
    r2 = channel_ranges(cols, signed)
    s, inv = le_solve(r1, r2, s_range, eps)
    a1 *= s[:, None]
    if bn_weight is not None:
        bn_weight *= s
    if bn_bias is not None:
        bn_bias *= s
    if b1 is not None:
        b1 *= s
    w2v *= inv.reshape(G, 1, gi, 1)
    return s


# --------------------------------------------------------------------------------------------
# a2  cross_layer_equalization (dfq.py:78-117)
# --------------------------------------------------------------------------------------------
def layer_absdiff_mean(w, w_prev):
    """float(torch.mean(torch.abs(W - W_prev))) (dfq.py:108).

    The reference's float32 mean has an implementation-defined summation order; oracle and HIP
    engine both define it as the float64 sum of the float32 |differences| divided by n and
    rounded once to float32 (order-independent to ~1e-16).
    """
    d = np.abs((w - w_prev).astype(F32)).astype(np.float64)
    return float(F32(d.sum() / d.size))


def cross_layer_equalization(spec, relations, s_range=(1e-8, 1e8), converge_thres=2e-7,
                             converge_count=20, signed=False, eps=0, max_sweeps=None,
                             trace=None):
    """Runs on a GraphSpec in place.  relations: list of (first_key, second_key, bn_key).

    ``max_sweeps`` (extension; None = reference behaviour) caps the number of sweeps.
    Returns (n_sweeps, S_cum) with S_cum[i] the cumulative scale vector of relation i
    (Relation.set_scale_vec, relation.py:20-24).
    """
    targ_keys = [k for k in spec.order if spec.nodes[k].kind == 'targ']
    S_cum = [None] * len(relations)
    diff = 10
    count = 0
    sweeps = 0
    while diff > converge_thres and count < converge_count:
        if max_sweeps is not None and sweeps >= max_sweeps:
            break
        prev = {k: spec.nodes[k].weight.copy() for k in targ_keys}
        for i, (kf, ks, kb) in enumerate(relations):
            nf, ns, nb = spec.nodes[kf], spec.nodes[ks], spec.nodes[kb]
            if nf.bias is None:                                       # dfq.py:91-92
                nf.bias = np.zeros(nf.weight.shape[0], dtype=F32)
            S = layer_equalization(nf.weight, ns.weight, nf.bias, nb.fake_weight, nb.fake_bias,
                                   s_range=s_range, signed=signed, eps=eps)
            S_cum[i] = S.copy() if S_cum[i] is None else (S_cum[i] * S).astype(F32)
        diff_tmp = 0.0
        for k in targ_keys:
            diff_tmp += layer_absdiff_mean(spec.nodes[k].weight, prev[k])
        if abs(diff - diff_tmp) > 1e-9:
            count = 0
            diff = diff_tmp
        else:
            count += 1
        sweeps += 1
        if trace is not None:
            trace.append(diff_tmp)
    return sweeps, S_cum


# --------------------------------------------------------------------------------------------
# a8  bias_absorption (dfq.py:121-164),  a9 clip_weight (dfq.py:167-170)
# --------------------------------------------------------------------------------------------
def _relu_between(spec, layer_second, layer_first):
    idx = layer_second
    while idx != layer_first:
        bots = spec.bottoms[idx]
        assert len(bots) == 1, 'graph in equalization relations should be 1-to-1 input-output'
        if spec.nodes[bots[0]].kind == 'relu':
            return True
        idx = bots[0]
    return False


def bias_absorption(spec, relations, N=3):
    for (kf, ks, kb) in relations:
        if not _relu_between(spec, ks, kf):
            continue
        nf, ns, nb = spec.nodes[kf], spec.nodes[ks], spec.nodes[kb]
        w2 = ns.weight
        O1 = nf.weight.shape[0]
        num_group = O1 // w2.shape[1]
        step_o = w2.shape[0] // num_group
        step_i = O1 // num_group
        c = (nb.fake_bias - (F32(N) * nb.fake_weight).astype(F32)).astype(F32)
        c = np.maximum(c, F32(0))
        wsum = _seq_sum_last(w2.reshape(w2.shape[0], w2.shape[1], -1))       # [O2, I2g]
        wc = np.zeros(w2.shape[0], dtype=F32)
        for g in range(num_group):
            wc[g * step_o:(g + 1) * step_o] = _matvec_f32(wsum[g * step_o:(g + 1) * step_o],
                                                          c[g * step_i:(g + 1) * step_i])
        for n in (nf, ns):
            if n.bias is None:
                n.bias = np.zeros(n.weight.shape[0], dtype=F32)
        nf.bias += -c
        nb.fake_bias += -c
        ns.bias += wc


def clip_weight(spec, range_clip=(-15, 15)):
    for k in spec.order:
        n = spec.nodes[k]
        if n.kind == 'targ':
            np.clip(n.weight, F32(range_clip[0]), F32(range_clip[1]), out=n.weight)


def _seq_sum_last(a3):
    acc = np.zeros(a3.shape[:-1], dtype=F32)
    for k in range(a3.shape[-1]):
        acc = (acc + a3[..., k]).astype(F32)
    return acc


def _matvec_f32(m, v):
    """Row-wise dot product; float64 accumulation rounded once to float32.

    The reference uses torch.matmul (BLAS sgemv, summation order unspecified); parity on the
    result is a 1e-5 contract, so the oracle takes the most accurate order-free definition.
    """
    return (m.astype(np.float64) @ v.astype(np.float64)).astype(F32)


# --------------------------------------------------------------------------------------------
# find_prev_bn (utils/layer_transform.py:299-344)
# --------------------------------------------------------------------------------------------
def find_prev_bn(spec, bn_seen, relu_attached, bot):
    """Breadth-first walk upwards from ``bot``; returns (bn_list, relu_list, connect_list).

    bn_list entries are (bn_key, branch_id_string).  ``bn_seen`` is the set of BN keys already
    registered by the caller (the reference passes a dict of modules).
    """
    frontier = [(b, str(i)) for i, b in enumerate(bot)]
    type_tmp = {str(i): 'one' for i in range(len(bot))}
    bn_list, relu_list, connect_list = [], [], []
    cat_add_found = False
    seen_targ_without_bn = set()
    while frontier:
        idx_bot, bid = frontier.pop(0)
        node = spec.nodes[idx_bot]
        if node.kind == 'op':                       # graph[idx] is a str in the reference
            name = str(idx_bot)
            if 'add' in name:
                if idx_bot in relu_attached:
                    type_tmp[bid] = 'add_{}'.format(relu_attached[idx_bot])
                else:
                    type_tmp[bid] = 'add'
                cat_add_found = True
            elif 'cat' in name:
                type_tmp[bid] = 'cat'
                cat_add_found = True
        elif (not cat_add_found) and node.kind == 'targ':
            assert bid[0] not in seen_targ_without_bn, \
                'Multiple conv/linear layer without batch_norm is not supported.'
            seen_targ_without_bn.add(bid[0])
        if idx_bot not in bn_seen:
            ups = spec.bottoms[idx_bot]
            frontier.extend([(u, bid + bid[0]) for u in ups])
            type_tmp[bid + bid[0]] = type_tmp[bid]
        else:
            bn_list.append((idx_bot, bid))
            relu_list.append(relu_attached[idx_bot])
            connect_list.append(type_tmp[bid])
    return bn_list, relu_list, connect_list


# --------------------------------------------------------------------------------------------
# a5  bias_correction (dfq.py:173-293)
# --------------------------------------------------------------------------------------------
_SQRT_2PI = math.sqrt(2.0 * math.pi)


def relu_mean(bn_weight, bn_bias):
    """calculate_mean of dfq.py:184: gamma*pdf(-beta/gamma) + beta*(1-cdf(-beta/gamma)).

    pdf/cdf are evaluated in float64 on the float32 ratio and rounded to float32 (scipy path of
    dfq.py:182-183); the surrounding arithmetic is float32.
    """
    from scipy.special import ndtr
    w = np.asarray(bn_weight, dtype=F32)
    b = np.asarray(bn_bias, dtype=F32)
    with np.errstate(divide='ignore', invalid='ignore'):
        t = ((-b) / w).astype(F32)
        t64 = t.astype(np.float64)
        pdf = (np.exp(-t64 ** 2 / 2.0) / _SQRT_2PI).astype(F32)
        cdf = ndtr(t64).astype(F32)
        e = ((w * pdf).astype(F32) + (b * (F32(1) - cdf).astype(F32)).astype(F32)).astype(F32)
    e = e.copy()
    e[e < 0] = 0                                    # NaN stays NaN, as in the reference
    return e


def bn_expectation(spec, entries):
    """Merge the per-BN expectations of one targ layer (dfq.py:221-278).

    entries: list of ((bn_key, bid), use_relu, connect_type) in find_prev_bn order.
    """
    branches = OrderedDict()
    for ent in entries:
        branches.setdefault(ent[0][1][0], []).append(ent)
    assert len(branches) == 1, 'Error while calculating expectation for bias correction'
    lst = sorted(list(branches.values())[0], key=lambda x: len(x[0][1]), reverse=True)

    def one(ent):
        (bn_key, _), use_relu, _ = ent
        n = spec.nodes[bn_key]
        if use_relu:
            return relu_mean(n.fake_weight, n.fake_bias)
        return n.fake_bias.copy()

    expect = one(lst[0])
    for ent in lst[1:]:
        e = one(ent)
        if ent[2] == 'cat':
            expect = np.concatenate([expect, e], 0)
        else:
            expect = (expect + e).astype(F32)
    return expect


def bias_correction(spec, signed=False, collect=None):
    """In place on a GraphSpec.  ``collect`` (optional dict) receives per-layer eps/expect/bias."""
    bn_seen = set()
    relu_attached = {}
    bias_prev = None
    for key in spec.order:
        bot = spec.bottoms[key]
        if bot is None or bot[0] == 'Data':
            continue
        node = spec.nodes[key]
        if node.kind == 'bn':
            bn_seen.add(key)
            relu_attached[key] = False
            if bias_prev is not None:
                node.fake_bias += bias_prev
                bias_prev = None
            continue
        if node.kind == 'relu':
            if bot[0] in bn_seen:
                relu_attached[bot[0]] = True
        if node.kind == 'targ':
            bn_list, relu_list, connect_list = find_prev_bn(spec, bn_seen, relu_attached, list(bot))
            eps = quant_error_rowsum(node.weight, signed=signed, num_bits=8)       # [O, I/g]
            entries = [(bn_list[i], relu_list[i], connect_list[i]) for i in range(len(bn_list))]
            expect = bn_expectation(spec, entries)
            num_group = expect.shape[0] // eps.shape[1]
            step_o = eps.shape[0] // num_group
            step_i = expect.shape[0] // num_group
            bias = np.zeros(eps.shape[0], dtype=F32)
            for g in range(num_group):
                bias[g * step_o:(g + 1) * step_o] = _matvec_f32(
                    eps[g * step_o:(g + 1) * step_o], expect[g * step_i:(g + 1) * step_i])
            if node.bias is None:
                node.bias = np.zeros(node.weight.shape[0], dtype=F32)
            node.bias += -bias
            bias_prev = -bias
            if collect is not None:
                collect[key] = dict(eps=eps, expect=expect, bias=bias)


# --------------------------------------------------------------------------------------------
# a7  merge_batchnorm (layer_transform.py:231-276),  a6 quantize_targ_layer (:279-296)
# --------------------------------------------------------------------------------------------
def merge_batchnorm(spec):
    for key in spec.order:
        bots = spec.bottoms[key]
        if bots is None:
            continue
        node = spec.nodes[key]
        for bk in bots:
            prev = spec.nodes[bk]
            if node.kind == 'bn' and prev.kind == 'targ' and node.gamma is not None:
                g = node.gamma
                k = (g / np.sqrt((node.var + F32(node.eps)).astype(F32))).astype(F32)
                w = prev.weight
                w *= k.reshape((-1,) + (1,) * (w.ndim - 1))
                if prev.bias is None:
                    prev.bias = np.zeros(w.shape[0], dtype=F32)
                shift = (node.beta - ((g * node.mean).astype(F32)
                                      / np.sqrt((node.var + F32(node.eps)).astype(F32))).astype(F32)).astype(F32)
                prev.bias[:] = ((prev.bias * k).astype(F32) + shift).astype(F32)
                node.fake_weight = np.abs(g).astype(F32)
                node.fake_bias = node.beta.copy()
                node.gamma = np.ones_like(g)
                node.var = np.ones_like(g)
                node.beta = np.zeros_like(g)
                node.mean = np.zeros_like(g)
                node.eps = 0.0
                break


def quantize_targ_layer(spec, bit_weight=8, bits_bias=16, return_codes=False):
    codes = {}
    for key in spec.order:
        n = spec.nodes[key]
        if n.kind != 'targ':
            continue
        w = n.weight
        out, q = uniform_quantize(w, bit_weight, float(w.min()), float(w.max()), return_codes=True)
        n.weight[...] = out
        codes[key] = q
        if n.bias is not None and bits_bias < 32:
            b = n.bias
            n.bias[...] = uniform_quantize(b, bits_bias, float(b.min()), float(b.max()))
    return codes if return_codes else None


# --------------------------------------------------------------------------------------------
# a12 create_relation (utils/relation.py:30-94)
# --------------------------------------------------------------------------------------------
_PASS_KINDS = ('bn', 'relu', 'qm', 'avgpool')


def create_relation(spec, delete_single=False):
    def find_prev(layer_idx, top_counter):
        bot = spec.bottoms[layer_idx]
        last_bn = None
        while len(bot) == 1 and bot[0] != 'Data' and top_counter[bot[0]] == 1:
            n = spec.nodes[bot[0]]
            if n.kind == 'bn':
                last_bn = bot[0]
            if n.kind == 'targ':
                return bot[0], last_bn
            elif not (n.kind in _PASS_KINDS or
                      (n.kind == 'op' and ('F.pad' in str(bot[0]) or 'torch.mean' in str(bot[0])))):
                return None, None
            bot = spec.bottoms[bot[0]]
        return None, None

    top_counter = {}
    for k in spec.order:
        if k == 'Data':
            continue
        for b in spec.bottoms[k]:
            top_counter[b] = top_counter.get(b, 0) + 1

    rel = OrderedDict()
    for k in spec.order:
        if spec.nodes[k].kind == 'targ':
            prev, bn = find_prev(k, top_counter)
            if prev in rel:
                rel.pop(prev)
            elif prev is not None:
                rel[prev] = (prev, k, bn)
    res = list(rel.values())
    if delete_single:
        groups = []
        for rr in res:
            gidx = -1
            for i, grp in enumerate(groups):
                for rp in grp:
                    if rr[0] == rp[1]:
                        gidx = i
                        break
            if gidx != -1:
                groups[gidx].append(rr)
            else:
                groups.append([rr])
        res = [rr for grp in groups if len(grp) > 1 for rr in grp]
    return res


# --------------------------------------------------------------------------------------------
# a10 QuantMeasure.forward (utils/quantize.py:102-119), eval mode
# --------------------------------------------------------------------------------------------
def sample_minmax_mean(x):
    """(mean_n min_chw x[n], mean_n max_chw x[n]) as float32 -- quantize.py:106-107."""
    x = np.asarray(x, dtype=F32)
    y = x.reshape(x.shape[0], -1)
    return _mean_f32(y.min(-1)), _mean_f32(y.max(-1))


def quant_measure_forward(x, running_min, running_max, num_bits=8, update_stat=False):
    """Eval-mode QuantMeasure.forward.  Returns (out, running_min, running_max)."""
    if update_stat:
        mn, mx = sample_minmax_mean(x)
        running_max = max(F32(running_max), mx)
        running_min = min(F32(running_min), mn)
    out = uniform_quantize(x, num_bits, float(running_min), float(running_max), num_chunks=16)
    return out, F32(running_min), F32(running_max)


# --------------------------------------------------------------------------------------------
# a11 merge_scale_to_weight (utils/quantize.py:145-174 conv, :269-289 linear)
# --------------------------------------------------------------------------------------------
def merge_scale_to_weight(weight, bias, scale=None, scale_prev=None, groups=1, linear=False):
    w = np.asarray(weight, dtype=F32)
    if scale_prev is not None:
        sp = np.asarray(scale_prev, dtype=F32).reshape(-1)
        if linear:
            w = (w * sp.reshape(1, -1)).astype(F32)          # QLinear multiplies (quantize.py:283)
        else:
            out = w.copy()
            step = w.shape[0] // groups
            step_s = w.shape[1]
            for g in range(groups):
                out[g * step:(g + 1) * step] = (w[g * step:(g + 1) * step]
                                                / sp[g * step_s:(g + 1) * step_s].reshape(1, -1, 1, 1)).astype(F32)
            w = out
    if scale is not None:
        s = np.asarray(scale, dtype=F32).reshape(-1)
        w = (w * s.reshape((-1,) + (1,) * (w.ndim - 1))).astype(F32)
        if bias is not None:
            bias = (np.asarray(bias, dtype=F32) * s).astype(F32)
    return w, bias


# --------------------------------------------------------------------------------------------
# f1  set_quant_minmax (utils/layer_transform.py:347-609): analytic activation ranges from the BN
#     proxies.  Restated for quant modules that belong to layers (`layer.quant`, the Q*Conv2d /
#     Q*Linear classes); tensor-op quant modules (replace_op machinery of the absent PyTransformer
#     submodule) are outside the restatement -- string nodes have no quant module here, which is
#     what the reference does when replace_op was not run.
# --------------------------------------------------------------------------------------------
def _pdf_cdf(x32):
    """scipy.stats.norm pdf / cdf of a float32 array, evaluated in float64, rounded to float32
    (standard_normal / standard_cdf of layer_transform.py:405-406)."""
    from scipy.special import ndtr
    x = np.asarray(x32, dtype=F32).astype(np.float64)
    with np.errstate(over='ignore', invalid='ignore'):
        pdf = (np.exp(-x ** 2 / 2.0) / _SQRT_2PI).astype(F32)
        cdf = ndtr(x).astype(F32)
    return pdf, cdf


def moments_relu(weight, bias):
    """calculate_mean / calculate_var of layer_transform.py:407-410 (float32 tensor arithmetic in the
    reference's operation order).  Returns (mean, var)."""
    w = np.asarray(weight, dtype=F32)
    b = np.asarray(bias, dtype=F32)
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        t = ((-b) / w).astype(F32)
        pdf, cdf = _pdf_cdf(t)
        mean = ((w * pdf).astype(F32) + (b * (F32(1) - cdf).astype(F32)).astype(F32)).astype(F32)
        # (1-cdf) * (b*b + w*w + mean*mean - 2*mean*b) + w*(b - 2*mean)*pdf + mean*mean*cdf
        poly = ((((b * b).astype(F32) + (w * w).astype(F32)).astype(F32) + (mean * mean).astype(F32)).astype(F32)
                - ((F32(2) * mean).astype(F32) * b).astype(F32)).astype(F32)
        t1 = ((F32(1) - cdf).astype(F32) * poly).astype(F32)
        t2 = ((w * (b - (F32(2) * mean).astype(F32)).astype(F32)).astype(F32) * pdf).astype(F32)
        t3 = ((mean * mean).astype(F32) * cdf).astype(F32)
        var = ((t1 + t2).astype(F32) + t3).astype(F32)
    return mean, var


def moments_relu6(weight, bias):
    """calculate_mean_6 / calculate_var_6 of layer_transform.py:411-418."""
    w = np.asarray(weight, dtype=F32)
    b = np.asarray(bias, dtype=F32)
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        lo = ((-b) / w).astype(F32)
        hi = ((F32(6) - b).astype(F32) / w).astype(F32)
        pdf_lo, cdf_lo = _pdf_cdf(lo)
        pdf_hi, cdf_hi = _pdf_cdf(hi)
        mean = (((w * (pdf_lo - pdf_hi).astype(F32)).astype(F32)
                 + (b * (cdf_hi - cdf_lo).astype(F32)).astype(F32)).astype(F32)
                + (F32(6) * (F32(1) - cdf_hi).astype(F32)).astype(F32)).astype(F32)
        poly = ((((b * b).astype(F32) + (w * w).astype(F32)).astype(F32) + (mean * mean).astype(F32)).astype(F32)
                - ((F32(2) * mean).astype(F32) * b).astype(F32)).astype(F32)
        t1 = ((cdf_hi - cdf_lo).astype(F32) * poly).astype(F32)
        t2 = ((w * F32(-6)).astype(F32) * pdf_hi).astype(F32)
        t3 = ((w * (b - (F32(2) * mean).astype(F32)).astype(F32)).astype(F32) * (pdf_lo - pdf_hi).astype(F32)).astype(F32)
        t4 = ((mean * mean).astype(F32) * cdf_lo).astype(F32)
        t5 = (((F32(6) - mean).astype(F32) ** 2).astype(F32) * (F32(1) - cdf_hi).astype(F32)).astype(F32)
        var = ((((t1 + t2).astype(F32) + t3).astype(F32) + t4).astype(F32) + t5).astype(F32)
    return mean, var


def bn_value_range(bias, weight, n):
    """(get_min_value, get_max_value) of layer_transform.py:403-404 as Python floats."""
    b = np.asarray(bias, dtype=F32)
    w = np.asarray(weight, dtype=F32)
    nw = (F32(n) * w).astype(F32)
    return float((b - nw).astype(F32).min()), float((b + nw).astype(F32).max())


def _clamped_range(bias, weight, n, relu):
    lo, hi = bn_value_range(bias, weight, n)
    if 'relu' in relu:                 # true for 'relu' and 'relu6'
        lo = max(0., lo)
    if 'relu6' in relu:
        hi = min(6., hi)
    return lo, hi


def set_quant_minmax(spec, is_detection=False, N=6, tensor_ops=None):
    """Returns OrderedDict key -> (running_min, running_max) for every layer (its input quantiser) and, for the
    tensor-op nodes listed in ``tensor_ops`` ({key: number of quantisers = inputs of the op}, the quantisers
    the reference keeps in its CustomTensorOP container, layer_transform.py:186-228), key -> [(min, max), ...]."""
    EPS = 1e-6
    tensor_ops = tensor_ops or {}
    bn_seen = OrderedDict()
    relu_attached = {}
    out = OrderedDict()

    def moments(w, b, relu):
        if relu == 'relu':
            return moments_relu(w, b)
        if relu == 'relu6':
            return moments_relu6(w, b)
        return b.copy(), (w * w).astype(F32)

    def branch_result(lst):
        """One branch (all entries hang off the same bottom): layer_transform.py:485-566."""
        lst = sorted(lst, key=lambda x: len(x[0][1]), reverse=True)
        (bn_key, bid), use_relu, connect_type = lst.pop(0)
        depth = len(bid)
        bias = spec.nodes[bn_key].fake_bias.copy()
        weight = spec.nodes[bn_key].fake_weight.copy()
        mean = var = None
        value_min = value_max = None
        if 'add' in connect_type:
            mean, var = moments(weight, bias, use_relu)
        else:
            value_min, value_max = _clamped_range(bias, weight, N, use_relu)
        while lst:
            bound = 0
            while bound < len(lst) and len(lst[bound][0][1]) == depth:
                bound += 1
            if bound == 0:
                depth = len(lst[0][0][1])       # cut depth
                continue
            for (bn_key, bid), relu_t, connect_type in lst[:bound]:
                bias = spec.nodes[bn_key].fake_bias.copy()
                weight = spec.nodes[bn_key].fake_weight.copy()
                if 'add' in connect_type:
                    m_t, v_t = moments(weight, bias, relu_t)
                    mean = (mean + m_t).astype(F32)
                    var = (var + v_t).astype(F32)
                    if 'relu6' in connect_type:
                        sd = np.sqrt((var + F32(EPS)).astype(F32)).astype(F32)
                        mean, var = moments_relu6(sd, mean)
                    elif 'relu' in connect_type:
                        sd = np.sqrt((var + F32(EPS)).astype(F32)).astype(F32)
                        mean, var = moments_relu(sd, mean)
                elif connect_type == 'cat':
                    lo, hi = _clamped_range(bias, weight, N, relu_t)
                    value_min = min(value_min, lo)
                    value_max = max(value_max, hi)
                else:
                    lo, hi = bn_value_range(bias, weight, N)
                    value_min += max(0., lo)        # `if use_relu_tmp` is always true (a non-empty string)
                    value_max += hi
            lst = lst[bound:]
            if connect_type == 'one':
                value_min /= (bound + 1)
                value_max /= (bound + 1)
        if 'add' in connect_type:
            sd = np.sqrt((var + F32(EPS)).astype(F32)).astype(F32)
            return bn_value_range(mean, sd, N)
        return value_min, value_max

    for key in spec.order:
        bot = spec.bottoms[key]
        if bot is None:
            continue
        node = spec.nodes[key]
        if node.kind == 'bn':
            bn_seen[key] = node
            relu_attached[key] = 'none'
            continue
        if node.kind == 'relu':
            relu_attached[bot[0]] = 'relu'
        elif node.kind == 'relu6':
            relu_attached[bot[0]] = 'relu6'
        if node.kind == 'targ':
            n_q = 1
        elif node.kind == 'op' and key in tensor_ops:
            n_q = int(tensor_ops[key])
        else:
            continue
        if len(bot) == 1 and bot[0] == 'Data':
            res = [(-1.0, 1.0) if is_detection else (-2.11790393, 2.64)]
        else:
            bn_list, relu_list, connect_list, no_bn = find_prev_bn_full(spec, bn_seen, relu_attached, list(bot))
            if n_q == len(bn_list):             # 1 to 1 mapping, quantiser i <- BN i
                res = []
                for (bn_key, bid), relu in zip(bn_list, relu_list):
                    bias = spec.nodes[bn_key].fake_bias.reshape(-1)
                    weight = spec.nodes[bn_key].fake_weight.reshape(-1)
                    if bid[0] in no_bn:         # case (d): a conv/linear between the BN and this node
                        lay = spec.nodes[no_bn[bid[0]]]
                        wsum = lay.weight.reshape(lay.weight.shape[0], lay.weight.shape[1], -1).astype(np.float64).sum(-1)
                        G = lay.groups
                        O, Ig = wsum.shape
                        go = O // G

                        def through(v):
                            v64 = v.astype(np.float64).reshape(G, Ig)
                            r = np.einsum('goi,gi->go', wsum.reshape(G, go, Ig), v64).reshape(-1)
                            return (r + lay.bias.astype(np.float64)).astype(F32)
                        res.append(bn_value_range(through(bias), through(weight), N))
                    else:
                        res.append(_clamped_range(bias, weight, N, relu))
            else:
                branches = OrderedDict()
                for ent, relu, ctype in zip(bn_list, relu_list, connect_list):
                    branches.setdefault(ent[1][0], []).append((ent, relu, ctype))
                per_branch = OrderedDict((b, branch_result(items)) for b, items in branches.items())
                if n_q == 1 and n_q < len(bn_list):                 # 1 to many
                    assert len(per_branch) == 1, 'Error occurs when setting min/max, should be 1 to many'
                    res = [list(per_branch.values())[0]]
                elif n_q < len(bn_list):                            # many to many
                    assert len(per_branch) == n_q, 'LENGTH NOT EQUAL {} vs {}'.format(len(per_branch), n_q)
                    res = [per_branch[str(i)] for i in range(n_q)]
                else:
                    raise AssertionError('Unknown error occured while setting min/max')
        out[key] = res[0] if node.kind == 'targ' else res
    return out


def find_prev_bn_full(spec, bn_seen, relu_attached, bot):
    """find_prev_bn (layer_transform.py:299-344) with the fourth return value: branch id ->
    key of the conv/linear layer found before the first BN (case d of set_quant_minmax)."""
    frontier = [(b, str(i)) for i, b in enumerate(bot)]
    type_tmp = {str(i): 'one' for i in range(len(bot))}
    bn_list, relu_list, connect_list = [], [], []
    no_bn = {}
    cat_add_found = False
    while frontier:
        idx_bot, bid = frontier.pop(0)
        node = spec.nodes[idx_bot]
        if node.kind == 'op':
            name = str(idx_bot)
            if 'add' in name:
                type_tmp[bid] = 'add_{}'.format(relu_attached[idx_bot]) if idx_bot in relu_attached else 'add'
                cat_add_found = True
            elif 'cat' in name:
                type_tmp[bid] = 'cat'
                cat_add_found = True
        elif (not cat_add_found) and node.kind == 'targ':
            assert bid[0] not in no_bn, 'Multiple conv/linear layer without batch_norm is not supported.'
            no_bn[bid[0]] = idx_bot
        if idx_bot not in bn_seen:
            frontier.extend([(u, bid + bid[0]) for u in spec.bottoms[idx_bot]])
            type_tmp[bid + bid[0]] = type_tmp[bid]
        else:
            bn_list.append((idx_bot, bid))
            relu_list.append(relu_attached[idx_bot])
            connect_list.append(type_tmp[bid])
    return bn_list, relu_list, connect_list, no_bn


# --------------------------------------------------------------------------------------------
# f3  the int8 calibration table of convert_ncnn.py:180-201 (weights block, then activations block)
# --------------------------------------------------------------------------------------------
def ncnn_table_lines(spec, act_ranges, names=None):
    """spec: GraphSpec; act_ranges: {targ key: (running_min, running_max)} as float32-representable
    Python floats.  Returns the lines the reference writes to model_int8_tensor.table."""
    keys = spec.targ_keys()
    if names is None:
        names = ['{}_param_0'.format(k) for k in keys] + [str(k) for k in keys]
    lines = []
    for i, k in enumerate(keys):
        w = spec.nodes[k].weight
        mi, ma = float(w.min()), float(w.max())
        scale = 128. / (max(abs(ma), abs(mi)))
        lines.append(' '.join([names[i]] + [str(scale)] * w.shape[0]))
    for i, k in enumerate(keys):
        mi, ma = float(F32(act_ranges[k][0])), float(F32(act_ranges[k][1]))
        scale = 128. / (max(abs(ma), abs(mi)))
        lines.append(' '.join([names[len(keys) + i], str(scale)]))
    return lines


# --------------------------------------------------------------------------------------------
# f3  ZeroQ's per-output-channel asymmetric weight quantiser
#     (ZeroQ/utils/quantization_utils/quant_utils.py:85-135 as called by quant_modules.py:161-171)
# --------------------------------------------------------------------------------------------
def zeroq_quant_rows(x, num_bits=8, return_codes=False):
    """x [O, ...]: per-row min/max, float32 tensor arithmetic in the reference's order."""
    x = np.asarray(x, dtype=F32)
    flat = x.reshape(x.shape[0], -1)
    mn = flat.min(1).astype(F32)
    mx = flat.max(1).astype(F32)
    n = F32(2 ** num_bits - 1)
    half = F32(2 ** (num_bits - 1))
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        span = np.maximum((mx - mn).astype(F32), F32(1e-8))
        # `n / tensor` with a Python int n is Tensor.__rtruediv__ = tensor.reciprocal() * n: two roundings
        scale = ((F32(1) / span).astype(F32) * n).astype(F32)
        zp = (np.rint((scale * mn).astype(F32)).astype(F32) + half).astype(F32)
        q = np.rint(((scale[:, None] * flat).astype(F32) - zp[:, None]).astype(F32)).astype(F32)
        q = np.clip(q, -half, half - F32(1)).astype(F32)
        y = ((q + zp[:, None]).astype(F32) / scale[:, None]).astype(F32)
    if return_codes:
        return y.reshape(x.shape), q.reshape(x.shape)
    return y.reshape(x.shape)


# --------------------------------------------------------------------------------------------
# f2  BatchNorm-statistics loss of ZeroQ's distillation (ZeroQ/distill_data.py:40-45, :172-196)
# --------------------------------------------------------------------------------------------
def bn_stat_losses(x, bn_mean, bn_std, eps=1e-6, denom=None):
    """x [N, C, H, W] -> (mean_loss, std_loss, dmean_loss/dx, dstd_loss/dx), evaluated in float64 (the reference
    computes in float32 with an unspecified reduction order; parity is a 1e-5 contract)."""
    x64 = np.asarray(x, dtype=F32).astype(np.float64)
    n, c = x64.shape[:2]
    flat = x64.reshape(n, c, -1)
    hw = flat.shape[-1]
    denom = float(c if denom is None else denom)
    e = (np.asarray(x, dtype=F32).reshape(n, c, -1) + F32(eps)).astype(F32).astype(np.float64)   # x + eps is rounded in float32
    mean = flat.mean(-1)
    me = e.mean(-1)
    std = np.sqrt(((e - me[..., None]) ** 2).sum(-1) / (hw - 1))
    bm = np.asarray(bn_mean, dtype=np.float64).reshape(1, c)
    bs = np.asarray(bn_std, dtype=np.float64).reshape(1, c)
    mean_loss = ((bm - mean) ** 2).sum() / denom
    std_loss = ((bs - std) ** 2).sum() / denom
    g_mean = np.broadcast_to((2.0 * (mean - bm) / (denom * hw))[..., None], flat.shape).reshape(x64.shape)
    g_std = ((2.0 * (std - bs) / denom)[..., None] * (e - me[..., None]) / ((hw - 1) * std[..., None])).reshape(x64.shape)
    return float(mean_loss), float(std_loss), g_mean, g_std
