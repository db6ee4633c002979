"""Fake-quantisation primitives with the call surface of the reference's ``utils/quantize.py``,
executed by the HIP engine (``dfq_fake_quant``, ``dfq_tensor_minmax``, ``dfq_sample_minmax_mean``).

  UniformQuantize / quantize   <- utils/quantize.py:14-87
  QuantMeasure                 <- utils/quantize.py:90-122
  QConv2d / QLinear (+ scale merging), Quant[N]Conv2d / Quant[N]Linear   <- :124-356
  set_layer_bits               <- :359-372

Numerics: the five float32 passes of quantize.py:70-74 are reproduced operation by operation in one
kernel (bit-exact codes); min/max that the reference turns into Python floats stay on the device
and feed the float64 scale recipe there, so there is no host round trip per call.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd.function import InplaceFunction

from .. import _ffi


# ------------------------------------------------------------------------------------------------
# engine calls on device tensors
# ------------------------------------------------------------------------------------------------
def _scratch(stage, n_words):
    return stage.new((n_words,), dtype=torch.int32)


def tensor_minmax(x_dev, stage=None):
    """Device float32[2] = (min, max) of a device tensor."""
    stage = stage or _ffi.Stage()
    out = stage.new((2,))
    scr = _scratch(stage, 2)
    _ffi.check(_ffi.lib().dfq_tensor_minmax(_ffi.ptr(x_dev), x_dev.numel(), _ffi.ptr(out), _ffi.ptr(scr),
                                            _ffi.stream_arg()))
    return out


def sample_minmax_mean(x_dev, n_rows, running=None, stage=None):
    """(mean over rows of row-min, mean over rows of row-max) as device float32[2]."""
    stage = stage or _ffi.Stage()
    out = stage.new((2,))
    scr = _scratch(stage, 2 * n_rows)
    _ffi.check(_ffi.lib().dfq_sample_minmax_mean(_ffi.ptr(x_dev), n_rows, x_dev.numel() // n_rows, _ffi.ptr(out),
                                                 _ffi.ptr(running), _ffi.ptr(scr), _ffi.stream_arg()))
    return out


def fake_quant_device(x_dev, out_dev, num_bits, symmetric, range_mode, min_value=0.0, max_value=0.0,
                      minmax_dev=None, codes_dev=None):
    _ffi.check(_ffi.lib().dfq_fake_quant(_ffi.ptr(x_dev), _ffi.ptr(out_dev), x_dev.numel(), int(num_bits),
                                         int(bool(symmetric)), int(range_mode), float(min_value), float(max_value),
                                         _ffi.ptr(minmax_dev), _ffi.ptr(codes_dev), _ffi.stream_arg()))


def _as_scalar_source(v):
    """min/max arguments may be Python numbers, 0-dim/1-element tensors or None."""
    if v is None:
        return None, None
    if isinstance(v, torch.Tensor):
        return None, v
    return float(v), None


def uniform_quantize(input, num_bits=8, min_value=None, max_value=None, inplace=False, symmetric=False,
                     num_chunks=None, return_codes=False):
    """UniformQuantize.forward (quantize.py:23-76)."""
    stage = _ffi.Stage()
    x = stage.bind(input)
    out = x if inplace else stage.new(x.shape)
    codes = stage.new(x.shape, dtype=torch.int32) if return_codes else None
    mn_f, mn_t = _as_scalar_source(min_value)
    mx_f, mx_t = _as_scalar_source(max_value)
    if mn_f is not None and mx_f is not None:
        # both are Python floats: float64 scale recipe on the host (quantize.py:49-66)
        fake_quant_device(x, out, num_bits, symmetric, 0, mn_f, mx_f, None, codes)
    else:
        # at least one bound is a tensor or None -> the reference's arithmetic becomes float32
        # tensor arithmetic (quantize.py:24-35); ranges are produced on the device
        B = x.shape[0] if x.dim() > 0 else 1
        nc = B if num_chunks is None else num_chunks
        rows = max(1, B // nc)
        mm = None
        if min_value is None or max_value is None:
            mm = sample_minmax_mean(x, rows, stage=stage)
        if min_value is None and max_value is None:
            pair = mm
        else:
            pair = stage.new((2,))
            for slot, (f, t) in enumerate(((mn_f, mn_t), (mx_f, mx_t))):
                if f is not None:
                    pair[slot] = f
                elif t is not None:
                    pair[slot] = stage.bind(t).reshape(-1)[0]
                else:
                    pair[slot] = mm[slot]
        fake_quant_device(x, out, num_bits, symmetric, 2, 0.0, 0.0, pair, codes)
    if inplace:
        stage.writeback()
        res = input
    else:
        res = stage.out_like(input, out)
    if return_codes:
        return res, stage.out_like(input, codes)
    return res


class UniformQuantize(InplaceFunction):
    """Straight-through fake quantiser; forward on the HIP engine (quantize.py:14-83)."""

    @staticmethod
    def forward(ctx, input, num_bits=8, min_value=None, max_value=None, inplace=False, symmetric=False,
                num_chunks=None):
        ctx.inplace = inplace
        ctx.num_bits = num_bits
        ctx.min_value = min_value
        ctx.max_value = max_value
        if inplace:
            ctx.mark_dirty(input)
        with torch.no_grad():
            return uniform_quantize(input, num_bits, min_value, max_value, inplace, symmetric, num_chunks)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output, None, None, None, None, None, None


def quantize(x, num_bits=8, min_value=None, max_value=None, inplace=False, symmetric=False, num_chunks=None):
    return UniformQuantize.apply(x, num_bits, min_value, max_value, inplace, symmetric, num_chunks)


# ------------------------------------------------------------------------------------------------
# QuantMeasure
# ------------------------------------------------------------------------------------------------
# QuantMeasure.forward with range tracking as ONE launch of persistent workgroups (dfq_quant_measure_fused) instead of two
# launches.  Opt-in: measured SLOWER on the MI355X (config 5: 3.73 vs 1.71 ms of QuantMeasure time per distilled batch; the largest
# activation 189 vs 148 us) -- the grid-wide meeting and the serialisation against other waiting kernels cost more than the
# launch boundary they replace.
_QM_FUSED = os.environ.get('DFQ_QM_FUSED', '0') == '1'


class QuantMeasure(nn.Module):
    """Activation range tracker + fake quantiser (quantize.py:90-122).

    Buffers ``running_min`` / ``running_max`` have shape [1] like the reference.  With
    ``update_stat`` the statistics kernels update a device-resident (min, max) pair in place and the
    quantiser reads its range from there: three launches, no host synchronisation.
    """

    def __init__(self, update_stat=False, num_bits=8, momentum=0.1):
        super().__init__()
        self.register_buffer('running_min', torch.zeros(1))
        self.register_buffer('running_max', torch.zeros(1))
        self.momentum = momentum
        self.num_bits = num_bits
        self.update_stat = update_stat

    def _packed_range(self, device):
        """running_min / running_max as the two halves of ONE device float32[2] (what the kernels update and read in place).
        The buffers are re-packed whenever something (``.to()``, ``load_state_dict`` of a fresh module, a caller assigning a
        new tensor) has separated them; packed, a forward pass needs no concatenation and no copy back."""
        mn, mx = self.running_min, self.running_max
        if (mn.device == device and mx.device == device and mn.dtype == torch.float32 and mx.dtype == torch.float32
                and mn.numel() == 1 and mx.numel() == 1 and mx.data_ptr() == mn.data_ptr() + 4
                and mn.untyped_storage().data_ptr() == mx.untyped_storage().data_ptr()):
            return torch.as_strided(mn, (2,), (1,))
        if mn.device != device or mx.device != device:
            return None                                   # a CPU-resident module: per-call shadow copies (below)
        pair = torch.empty(2, dtype=torch.float32, device=device)
        pair[0:1].copy_(mn.reshape(1))
        pair[1:2].copy_(mx.reshape(1))
        self._buffers['running_min'] = pair[0:1]
        self._buffers['running_max'] = pair[1:2]
        return pair

    def forward(self, input):
        with torch.no_grad():
            stage = _ffi.Stage()
            x = stage.bind(input)
            n = x.shape[0]
            running = self._packed_range(stage.device)
            shadow = running is None
            if self.update_stat and not self.training and not shadow and x.numel() > 0:
                # the common calibration path (improve_dfq.py:280-297): two launches, no temporaries besides the output
                # (DFQ_QM_FUSED=1: one launch of persistent workgroups -- measured slower, see _QM_FUSED)
                lib = _ffi.lib()
                fused = _QM_FUSED
                words = 4 * n + (4 if fused else 0)
                sc = getattr(self, '_qm_scratch', None)
                if sc is None or sc.numel() != words or sc.device != x.device:
                    sc = torch.zeros(words, dtype=torch.int32, device=x.device)
                    self._qm_scratch, self._qm_parity, self._qm_arrivals = sc, 0, 0
                out = stage.new(x.shape)
                if fused:
                    _ffi.check(lib.dfq_quant_measure_fused(_ffi.ptr(x), _ffi.ptr(out), n, x.numel() // n, int(self.num_bits),
                                                           _ffi.ptr(running), _ffi.ptr(sc), self._qm_parity, self._qm_arrivals,
                                                           _ffi.stream_arg()))
                    self._qm_arrivals += int(lib.dfq_quant_measure_fused_grid(n, x.numel() // n))
                else:
                    _ffi.check(lib.dfq_quant_measure(_ffi.ptr(x), _ffi.ptr(out), n, x.numel() // n, int(self.num_bits),
                                                     _ffi.ptr(running), _ffi.ptr(sc), self._qm_parity, _ffi.stream_arg()))
                self._qm_parity ^= 1
            else:
                if shadow:
                    running = torch.cat([stage.bind(self.running_min).reshape(1), stage.bind(self.running_max).reshape(1)])
                if self.update_stat:
                    sample_minmax_mean(x, n, running=running, stage=stage)       # quantize.py:106-107
                pair = running                                                    # eval: running range
                if self.training:
                    pair = sample_minmax_mean(x, n, stage=stage)                  # quantize.py:109-113
                    running.mul_(1 - self.momentum).add_(pair * self.momentum)
                if shadow:
                    self.running_min.copy_(running[0:1])
                    self.running_max.copy_(running[1:2])
                out = stage.new(x.shape)
                # float(min_value), float(max_value) -> float64 recipe, evaluated on the device
                fake_quant_device(x, out, self.num_bits, False, 1, 0.0, 0.0, pair, None)
            out = stage.out_like(input, out)              # a CPU input gets a CPU result, whichever path ran
        if input.requires_grad:
            out = input + (out - input).detach()          # straight-through estimator (quantize.py:79-83), OUTSIDE no_grad
        return out

    def set_update_stat(self, update_stat):
        self.check_fused_status()
        self.update_stat = update_stat

    def check_fused_status(self):
        """DFQ_QM_FUSED=1 only: the one-launch variant cannot report an abandoned grid-wide wait from inside forward() without a
        synchronisation per call, so the calibration loop's own boundary does it -- `set_update_stat` (improve_dfq.py:299-309 calls
        it before and after `update_quant_range`).  After an abandon the outputs and the range of the failed call are undefined:
        raises, and the scratch (whose error word is sticky) is dropped so that the module can be used again."""
        sc = getattr(self, '_qm_scratch', None)
        if not _QM_FUSED or sc is None or sc.device.type != 'cuda':
            return
        n = (sc.numel() - 4) // 4
        rc = _ffi.lib().dfq_quant_measure_fused_status(_ffi.ptr(sc), n, _ffi.stream_arg())
        if rc != 0:
            self._qm_scratch = None
            _ffi.check(rc)


# ------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------
def _quant_weight(weight, num_bits):
    """quantize(w, bits, float(w.min()), float(w.max())) without the host round trip."""
    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, w):
            with torch.no_grad():
                stage = _ffi.Stage()
                x = stage.bind(w)
                mm = tensor_minmax(x, stage)
                out = stage.new(x.shape)
                fake_quant_device(x, out, num_bits, False, 1, 0.0, 0.0, mm, None)
                return stage.out_like(w, out)

        @staticmethod
        def backward(ctx, g):
            return g
    return _Fn.apply(weight)


class _ScaleMixin:
    """QConv2d/QLinear scale handling (quantize.py:136-174, :260-289)."""

    def set_scale(self, scale=None, scale_prev=None):
        if scale is not None:
            shape = (-1, 1, 1, 1) if isinstance(self, nn.Conv2d) else (-1, 1)
            self.register_parameter('scale', nn.Parameter(scale.view(*shape)))
        if scale_prev is not None:
            self.scale_prev = scale_prev

    def merge_scale_to_weight(self):
        from ..improve_dfq import merge_scale_into_layer
        merge_scale_into_layer(self)


class QConv2d(nn.Conv2d, _ScaleMixin):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, num_bits=8, num_bits_act=8, num_bits_bias=16, momentum=0.1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.num_bits = num_bits
        self.num_bits_bias = num_bits_bias
        self.quant = QuantMeasure(num_bits=num_bits_act, momentum=momentum)

    def merge_scale_prev(self, weight, scale_prev):
        step_s = weight.shape[1]
        sp = scale_prev.reshape(-1)
        per_row = sp.view(self.groups, step_s).repeat_interleave(weight.shape[0] // self.groups, dim=0)
        return weight / per_row.view(weight.shape[0], step_s, 1, 1)

    def merge_scale(self, weight, bias, scale):
        weight = weight * scale
        if bias is not None:
            bias = bias * scale.view(-1)
        return weight, bias

    def forward(self, input):
        input = self.quant(input)
        sweight, sbias = self.weight, self.bias
        if getattr(self, 'scale_prev', None) is not None:
            sweight = self.merge_scale_prev(sweight, self.scale_prev)
        if getattr(self, 'scale', None) is not None:
            sweight, sbias = self.merge_scale(sweight, sbias, self.scale)
        qweight = _quant_weight(sweight, self.num_bits)
        qbias = quantize(sbias, num_bits=self.num_bits_bias) if sbias is not None else None
        return F.conv2d(input, qweight, qbias, self.stride, self.padding, self.dilation, self.groups)


class QuantConv2d(nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, num_bits=8, num_bits_act=8, num_bits_bias=16, momentum=0.1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.num_bits = num_bits
        self.num_bits_bias = num_bits_bias
        self.quant = QuantMeasure(num_bits=num_bits_act, momentum=momentum)

    def forward(self, input):
        input = self.quant(input)
        qweight = _quant_weight(self.weight, self.num_bits)
        qbias = quantize(self.bias, num_bits=self.num_bits_bias) if self.bias is not None else None
        return F.conv2d(input, qweight, qbias, self.stride, self.padding, self.dilation, self.groups)


class QuantNConv2d(nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, num_bits=8, num_bits_act=8, momentum=0.1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.quant = QuantMeasure(num_bits=num_bits_act, momentum=momentum)

    def forward(self, input):
        input = self.quant(input)
        return F.conv2d(input, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)


class QLinear(nn.Linear, _ScaleMixin):
    def __init__(self, in_features, out_features, bias=True, num_bits=8, num_bits_act=8, num_bits_bias=16,
                 momentum=0.1):
        super().__init__(in_features, out_features, bias)
        self.num_bits = num_bits
        self.num_bits_bias = num_bits_bias
        self.quant = QuantMeasure(num_bits=num_bits_act, momentum=momentum)

    def merge_scale_prev(self, weight, scale_prev):
        return weight * scale_prev.view(1, -1)            # quantize.py:282-283 (multiplies)

    def merge_scale(self, weight, bias, scale):
        weight = weight * scale
        if bias is not None:
            bias = bias * scale.view(-1)
        return weight, bias

    def forward(self, input):
        input = self.quant(input)
        sweight, sbias = self.weight, self.bias
        if getattr(self, 'scale_prev', None) is not None:
            sweight = self.merge_scale_prev(sweight, self.scale_prev)
        if getattr(self, 'scale', None) is not None:
            sweight, sbias = self.merge_scale(sweight, sbias, self.scale)
        qweight = _quant_weight(sweight, self.num_bits)
        qbias = quantize(sbias, num_bits=self.num_bits_bias) if sbias is not None else None
        return F.linear(input, qweight, qbias)


class QuantLinear(nn.Linear):
    def __init__(self, in_features, out_features, bias=True, num_bits=8, num_bits_act=8, num_bits_bias=16,
                 momentum=0.1):
        super().__init__(in_features, out_features, bias)
        self.num_bits = num_bits
        self.num_bits_bias = num_bits_bias
        self.quant = QuantMeasure(num_bits=num_bits_act, momentum=momentum)

    def forward(self, input):
        input = self.quant(input)
        qweight = _quant_weight(self.weight, self.num_bits)
        qbias = quantize(self.bias, num_bits=self.num_bits_bias) if self.bias is not None else None
        return F.linear(input, qweight, qbias)


class QuantNLinear(nn.Linear):
    def __init__(self, in_features, out_features, bias=True, num_bits=8, num_bits_act=8, momentum=0.1):
        super().__init__(in_features, out_features, bias)
        self.quant = QuantMeasure(num_bits=num_bits_act, momentum=momentum)

    def forward(self, input):
        input = self.quant(input)
        return F.linear(input, self.weight, self.bias)


def set_layer_bits(graph, bits_weight=8, bits_activation=8, bits_bias=16, targ_type=None):
    print("Setting num_bits for targ layers...")
    assert targ_type != None, "targ_type cannot be None"
    for idx in graph:
        if type(graph[idx]) in targ_type:
            if hasattr(graph[idx], 'quant'):
                graph[idx].quant = QuantMeasure(bits_activation)
            if hasattr(graph[idx], 'num_bits'):
                graph[idx].num_bits = bits_weight
            if hasattr(graph[idx], 'num_bits_bias'):
                graph[idx].num_bits_bias = bits_bias
