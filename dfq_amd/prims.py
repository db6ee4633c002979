"""Stand-alone steps of the calibration path (C ABI section "Stand-alone steps").

Thin ctypes wrappers: every function stages its tensor arguments on the GPU (in place when they already
live there), calls ONE entry point of libdfq_hip.so and returns device-side results moved back to the
caller's device.  The plans in dfq_amd.dfq are the fast path; these exist so that a caller can drive the
reference's algorithm step by step (dfq.py:28-75, :105-108, :281-287) and for the per-channel quantiser.
"""
from __future__ import annotations

import torch

from . import _ffi


def _weight_geometry(w):
    out_ch = w.shape[0]
    in_per_group = w.shape[1]
    khkw = w[0, 0].numel() if w.dim() > 2 else 1
    return out_ch, in_per_group, khkw


def row_range(weight, signed=False):
    """range of every output row of `weight` (dfq.py:50-55 on the first layer) -> float32 [O]."""
    lib = _ffi.lib()
    with torch.no_grad():
        stage = _ffi.Stage()
        w = stage.bind(weight)
        out = stage.new((w.shape[0],))
        _ffi.check(lib.dfq_row_range(_ffi.ptr(w), w.shape[0], w[0].numel(), int(bool(signed)), _ffi.ptr(out),
                                     _ffi.stream_arg()))
        return stage.out_like(weight, out)


def col_range(weight_second, first_out_channels, signed=False):
    """range of every paired input channel of the second layer (the view of dfq.py:41-46) -> float32 [O1]."""
    lib = _ffi.lib()
    with torch.no_grad():
        stage = _ffi.Stage()
        w = stage.bind(weight_second)
        o2, i2g, khkw = _weight_geometry(w)
        groups = first_out_channels // i2g if first_out_channels != i2g else 1
        out = stage.new((groups * i2g,))
        _ffi.check(lib.dfq_col_range(_ffi.ptr(w), o2, i2g, khkw, groups, int(bool(signed)), _ffi.ptr(out),
                                     _ffi.stream_arg()))
        return stage.out_like(weight_second, out)


def le_solve(r1, r2, s_range=(1e-8, 1e8), eps=0):
    """(S, 1/S as the reference applies it) from the two range vectors (dfq.py:58-59, :73)."""
    lib = _ffi.lib()
    with torch.no_grad():
        stage = _ffi.Stage()
        a, b = stage.bind(r1), stage.bind(r2)
        s, inv = stage.new(a.shape), stage.new(a.shape)
        _ffi.check(lib.dfq_le_solve(_ffi.ptr(a), _ffi.ptr(b), a.numel(), float(eps), float(s_range[0]), float(s_range[1]),
                                    _ffi.ptr(s), _ffi.ptr(inv), _ffi.stream_arg()))
        return stage.out_like(r1, s), stage.out_like(r1, inv)


def le_apply(weight_first, weight_second, bias_first, bn_weight, bn_bias, S, Sinv):
    """In place: W1 rows, b1, BN proxies *= S; W2 input channels *= Sinv (dfq.py:62-73)."""
    lib = _ffi.lib()
    with torch.no_grad():
        stage = _ffi.Stage()
        w1, w2 = stage.bind(weight_first), stage.bind(weight_second)
        o2, i2g, khkw = _weight_geometry(w2)
        _ffi.check(lib.dfq_le_apply(_ffi.ptr(w1), w1.shape[0], w1[0].numel(), _ffi.ptr(w2), o2, i2g, khkw,
                                    _ffi.ptr(stage.bind(bias_first)), _ffi.ptr(stage.bind(bn_weight)),
                                    _ffi.ptr(stage.bind(bn_bias)), _ffi.ptr(stage.bind(S)), _ffi.ptr(stage.bind(Sinv)),
                                    _ffi.stream_arg()))
        stage.writeback()


def le_pair(weight_first, weight_second, bias_first, bn_weight=None, bn_bias=None, s_range=(1e-8, 1e8),
            signed=False, eps=0):
    """_layer_equalization (dfq.py:28-75) through the stand-alone steps; returns S."""
    lib = _ffi.lib()
    with torch.no_grad():
        stage = _ffi.Stage()
        w1, w2 = stage.bind(weight_first), stage.bind(weight_second)
        o2, i2g, khkw = _weight_geometry(w2)
        o1 = w1.shape[0]
        S = stage.new((o1,))
        work = stage.new((3 * o1,))
        _ffi.check(lib.dfq_le_pair(_ffi.ptr(w1), o1, w1[0].numel(), _ffi.ptr(w2), o2, i2g, khkw,
                                   _ffi.ptr(stage.bind(bias_first)), _ffi.ptr(stage.bind(bn_weight)),
                                   _ffi.ptr(stage.bind(bn_bias)), float(s_range[0]), float(s_range[1]),
                                   int(bool(signed)), float(eps), _ffi.ptr(S), _ffi.ptr(work), _ffi.stream_arg()))
        stage.writeback()
        return stage.out_like(weight_first, S)


def absdiff_mean(weight, prev):
    """float(torch.mean(torch.abs(weight - prev))) of dfq.py:108 as a Python float."""
    lib = _ffi.lib()
    with torch.no_grad():
        stage = _ffi.Stage()
        a, b = stage.bind(weight), stage.bind(prev)
        n = a.numel()
        out = stage.new((1,))
        scratch = stage.new((int(lib.dfq_absdiff_mean_scratch_bytes(n)) // 8 + 1,), dtype=torch.float64)
        _ffi.check(lib.dfq_absdiff_mean(_ffi.ptr(a), _ffi.ptr(b), n, _ffi.ptr(out), _ffi.ptr(scratch), _ffi.stream_arg()))
        return float(out.item())


def fake_quant_rows(x, num_bits=8, min_values=None, max_values=None, symmetric=False, return_codes=False):
    """Per-output-channel fake-quant: row r of `x` (first dimension) with its own (min, max)."""
    lib = _ffi.lib()
    with torch.no_grad():
        stage = _ffi.Stage()
        xx = stage.bind(x)
        rows, row_len = xx.shape[0], xx[0].numel()
        y = stage.new(xx.shape)
        codes = stage.new(xx.shape) if return_codes else None
        mm = stage.new((rows, 2))
        _ffi.check(lib.dfq_fake_quant_rows(_ffi.ptr(xx), _ffi.ptr(y), rows, row_len, _ffi.ptr(stage.bind(min_values)),
                                           _ffi.ptr(stage.bind(max_values)), int(num_bits), int(bool(symmetric)),
                                           _ffi.ptr(codes), _ffi.ptr(mm), _ffi.stream_arg()))
        res = stage.out_like(x, y)
        if return_codes:
            return res, stage.out_like(x, codes), stage.out_like(x, mm)
        return res


def zeroq_quant_rows(x, num_bits=8, min_values=None, max_values=None, return_codes=False):
    """ZeroQ's per-output-channel asymmetric quantiser (quant_utils.py:85-135 via quant_modules.py:161-171)."""
    lib = _ffi.lib()
    with torch.no_grad():
        stage = _ffi.Stage()
        xx = stage.bind(x)
        rows, row_len = xx.shape[0], xx[0].numel()
        y = stage.new(xx.shape)
        codes = stage.new(xx.shape) if return_codes else None
        _ffi.check(lib.dfq_zeroq_quant_rows(_ffi.ptr(xx), _ffi.ptr(y), rows, row_len, _ffi.ptr(stage.bind(min_values)),
                                            _ffi.ptr(stage.bind(max_values)), int(num_bits), _ffi.ptr(codes), None,
                                            _ffi.stream_arg()))
        res = stage.out_like(x, y)
        return (res, stage.out_like(x, codes)) if return_codes else res


def grouped_matvec(eps, expect, groups=1):
    """bias[o] = eps[o, :] . expect[group(o)] (dfq.py:281-287) -> float32 [O]."""
    lib = _ffi.lib()
    with torch.no_grad():
        stage = _ffi.Stage()
        e, x = stage.bind(eps), stage.bind(expect)
        out = stage.new((e.shape[0],))
        _ffi.check(lib.dfq_grouped_matvec(_ffi.ptr(e), _ffi.ptr(x), e.shape[0], e.shape[1], int(groups), _ffi.ptr(out),
                                          _ffi.stream_arg()))
        return stage.out_like(eps, out)
