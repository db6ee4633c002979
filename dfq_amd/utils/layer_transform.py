"""Graph-level passes around the equalisation/correction core, with the call surface of the
reference's ``utils/layer_transform.py``:

  merge_batchnorm      <- utils/layer_transform.py:231-276   (engine: dfq_fold_batchnorm)
  quantize_targ_layer  <- utils/layer_transform.py:279-296   (engine: dfq_quant_plan_*)
  find_prev_bn         <- utils/layer_transform.py:299-344   (host graph walk, O(#nodes))

The graph model is the reference's: ``graph`` maps key -> nn.Module | str (tensor ops are strings whose
key contains 'add' / 'cat' / ...), ``bottoms`` maps key -> list of input keys | None.
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn

from .. import _ffi
from .quantize import QConv2d, QuantConv2d, QuantNConv2d, QLinear, QuantLinear, QuantNLinear

_CONV_TYPES = (nn.Conv2d, QConv2d, QuantConv2d, QuantNConv2d)
_LINEAR_TYPES = (nn.Linear, QLinear, QuantLinear, QuantNLinear)


def _ensure_bias(layer):
    """The reference adds a zero bias Parameter when a layer has none (layer_transform.py:253-254)."""
    if layer.bias is None:
        layer.bias = nn.Parameter(torch.zeros(layer.weight.size(0), dtype=torch.float32,
                                              device=layer.weight.device), requires_grad=False)
    return layer.bias


def merge_batchnorm(model, graph, bottoms, targ_type=[QConv2d]):
    """Fold every BatchNorm2d that directly follows a targ layer into that layer.

    W <- W * gamma/sqrt(var+eps) per output channel, b <- b*gamma/sqrt(var+eps) + beta -
    gamma*mean/sqrt(var+eps); the BN keeps ``fake_weight = |gamma|`` and ``fake_bias = beta`` for the
    later passes and becomes an identity (eps = 0).
    """
    lib = _ffi.lib()
    with torch.no_grad():
        stage = _ffi.Stage()
        for key in graph:
            bots = bottoms[key]
            if bots is None:
                continue
            bn = graph[key]
            if type(bn) != nn.BatchNorm2d:
                continue
            for bk in bots:
                layer = graph[bk]
                if type(layer) not in targ_type:
                    continue
                _ensure_bias(layer)
                w = stage.bind(layer.weight)
                b = stage.bind(layer.bias)
                gamma, beta = stage.bind(bn.weight), stage.bind(bn.bias)
                mean, var = stage.bind(bn.running_mean), stage.bind(bn.running_var)
                fw = torch.empty_like(gamma)
                fb = torch.empty_like(gamma)
                _ffi.check(lib.dfq_fold_batchnorm(_ffi.ptr(w), _ffi.ptr(b), w.shape[0], w[0].numel(), _ffi.ptr(gamma),
                                                  _ffi.ptr(beta), _ffi.ptr(mean), _ffi.ptr(var),
                                                  ctypes.c_float(bn.eps), _ffi.ptr(fw), _ffi.ptr(fb),
                                                  _ffi.stream_arg()))
                bn.register_buffer('fake_weight', stage.out_like(bn.weight, fw))
                bn.register_buffer('fake_bias', stage.out_like(bn.weight, fb))
                bn.eps = 0
                break
        stage.writeback()
    return model


def quantize_targ_layer(graph, bit_weight=8, bits_bias=16, targ_type=None, return_codes=False):
    """Per-tensor asymmetric fake-quant of every targ layer's weight (and bias unless 32 bit).

    Two launches for the whole network: one multi-tensor min/max, one multi-tensor quantise.
    ``return_codes`` (extension) additionally returns {key: int32 code tensor of the weight}.
    """
    print("Quantizing Layer parameters")
    if bits_bias == 32:
        print("Skipping bias quantization (32 bits)")
    assert targ_type != None, "targ_type cannot be None!"
    lib = _ffi.lib()
    with torch.no_grad():
        stage = _ffi.Stage()
        segs, keep, codes = [], [], {}
        for key in graph:
            layer = graph[key]
            if type(layer) not in targ_type:
                continue
            w = stage.bind(layer.weight)
            c = None
            if return_codes:
                c = stage.new(w.shape, dtype=torch.int32)
                codes[key] = c
            segs.append(_ffi.DfqSegment(w.data_ptr(), w.numel(), int(bit_weight), 0, c.data_ptr() if c is not None else None))
            keep.append(w)
            if layer.bias is not None and bits_bias < 32:
                b = stage.bind(layer.bias)
                segs.append(_ffi.DfqSegment(b.data_ptr(), b.numel(), int(bits_bias), 0, None))
                keep.append(b)
        if segs:
            arr = (_ffi.DfqSegment * len(segs))(*segs)
            plan = ctypes.c_void_p()
            _ffi.check(lib.dfq_quant_plan_create(arr, len(segs), ctypes.byref(plan)))
            try:
                _ffi.check(lib.dfq_quant_plan_run(plan, _ffi.stream_arg()))
                _ffi.synchronize()
            finally:
                lib.dfq_quant_plan_destroy(plan)
        stage.writeback()
    if return_codes:
        return graph, codes
    return graph


def find_prev_bn(bn_module, relu_attached, graph, bottoms, bot):
    """Breadth-first walk upwards from ``bot`` to the nearest BatchNorm on every path.

    Returns (bn_list, relu_attach_list, connect_type_list, targ_without_bn) like
    layer_transform.py:299-344: ``bn_list`` holds (bn module, branch id) where the branch id is a
    string of length = depth of the hit; the connect type ('one' / 'add' / 'add_<flag>' / 'cat')
    is inherited from the last add/cat node passed on the way up.
    """
    frontier = [(b, str(i)) for i, b in enumerate(bot)]
    ctype = {str(i): 'one' for i in range(len(bot))}
    bn_list, relu_list, connect_list = [], [], []
    no_bn_targets = {}
    merged = False
    while frontier:
        key, bid = frontier.pop(0)
        node = graph[key]
        if isinstance(node, str):
            if 'add' in key:
                ctype[bid] = 'add_{}'.format(relu_attached[key]) if key in relu_attached else 'add'
                merged = True
            elif 'cat' in key:
                ctype[bid] = 'cat'
                merged = True
        elif not merged and type(node) in _CONV_TYPES + _LINEAR_TYPES:
            print("Warning: {} layer before first batch norm layer detected. The calculated value range might be off.".format(type(node)))
            assert bid[0] not in no_bn_targets, "Multiple conv/linear layer without batch_norm is not supported."
            no_bn_targets[bid[0]] = ("conv" if type(node) in _CONV_TYPES else "linear", node)
        if key in bn_module:
            bn_list.append((bn_module[key], bid))
            relu_list.append(relu_attached[key])
            connect_list.append(ctype[bid])
        else:
            deeper = bid + bid[0]
            frontier.extend((up, deeper) for up in bottoms[key])
            ctype[deeper] = ctype[bid]
    return bn_list, relu_list, connect_list, no_bn_targets
