"""Scale folding and distilled-range calibration with the call surface of the reference's
``improve_dfq.py``:

  transform_quant_layer   <- improve_dfq.py:144-172  (+ utils/quantize.py:145-174, :269-289)
  update_quant_range      <- improve_dfq.py:280-297
  set_update_stat         <- improve_dfq.py:299-309

The abandoned experiments of that file (GradHook, update_scale, bias_correction_distill, ...; call
sites commented out in main_cls.py:157-175,192-194) are out of scope.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _ffi
from .utils.quantize import (QConv2d, QLinear, QuantConv2d, QuantLinear, QuantMeasure, QuantNConv2d,
                             QuantNLinear)


def merge_scale_into_layer(layer):
    """QConv2d/QLinear.merge_scale_to_weight (quantize.py:145-156, :269-280) on the engine.

    scale_prev: W[o, i, :] /= scale_prev[g(o)*I/g + i]  for convs (quantize.py:158-167),
                W[o, i]    *= scale_prev[i]             for linears (quantize.py:282-283);
    scale:      W[o, ...]  *= scale[o], b[o] *= scale[o] (quantize.py:169-174).
    """
    lib = _ffi.lib()
    with torch.no_grad():
        stage = _ffi.Stage()
        w = stage.bind(layer.weight)
        is_conv = w.dim() == 4
        khkw = w[0, 0].numel() if is_conv else 1
        sp = getattr(layer, 'scale_prev', None)
        if sp is not None:
            spd = stage.bind(sp.detach().reshape(-1).contiguous())
            groups = layer.groups if is_conv else 1
            _ffi.check(lib.dfq_scale_cols(_ffi.ptr(w), w.shape[0], w.shape[1], khkw, groups, _ffi.ptr(spd),
                                          1 if is_conv else 0, _ffi.stream_arg()))
            layer.scale_prev = None
        sc = getattr(layer, 'scale', None)
        if sc is not None:
            scd = stage.bind(sc.detach().reshape(-1).contiguous())
            _ffi.check(lib.dfq_scale_rows(_ffi.ptr(w), w.shape[0], w[0].numel(), _ffi.ptr(scd), 0, _ffi.stream_arg()))
            if layer.bias is not None:
                b = stage.bind(layer.bias)
                _ffi.check(lib.dfq_vec_op(_ffi.ptr(b), _ffi.ptr(scd), b.numel(), 0, _ffi.stream_arg()))
            layer.scale = None
        stage.writeback()


def _swap_modules(model, mapping):
    """Replace every module whose exact type is a key of `mapping`, carrying parameters and the
    activation quantiser over (what the reference gets from TorchTransformer.trans_layers)."""
    swapped = {}
    for name, child in list(model.named_children()):
        new_cls = mapping.get(type(child))
        if new_cls is None:
            swapped.update(_swap_modules(child, mapping))
            continue
        if isinstance(child, nn.Conv2d):
            new = new_cls(child.in_channels, child.out_channels, child.kernel_size, child.stride, child.padding,
                          child.dilation, child.groups, child.bias is not None)
        else:
            new = new_cls(child.in_features, child.out_features, child.bias is not None)
        new.weight = child.weight
        new.bias = child.bias
        if hasattr(child, 'quant'):
            new.quant = child.quant
        for attr in ('num_bits', 'num_bits_bias'):
            if hasattr(child, attr) and hasattr(new, attr):
                setattr(new, attr, getattr(child, attr))
        new.train(child.training)
        setattr(model, name, new)
        swapped[child] = new
    return swapped


def transform_quant_layer(model, graph, res, trainable=False):
    """Fold the learned/equalisation scales into the weights of every relation's layers, drop the
    scale attributes and swap QConv2d/QLinear for their plain quantised counterparts."""
    for rr in res:
        layer_first, layer_second, _ = rr.get_idxs()
        for key in (layer_first, layer_second):
            layer = graph[key]
            merge_scale_into_layer(layer)
            for attr in ('scale', 'scale_prev'):
                if hasattr(layer, attr):
                    if attr in layer._parameters:
                        del layer._parameters[attr]
                    elif attr in layer.__dict__:
                        delattr(layer, attr)
    if trainable:
        mapping = {QConv2d: QuantConv2d, QLinear: QuantLinear}
    else:
        mapping = {QConv2d: QuantNConv2d, QLinear: QuantNLinear}
    swapped = _swap_modules(model, mapping)
    for key in graph:
        if not isinstance(graph[key], str) and graph[key] in swapped:
            graph[key] = swapped[graph[key]]
    return model


def set_update_stat(model, targ_type, update_stat):
    """Toggle ``update_stat`` on every module whose type is in ``targ_type`` (improve_dfq.py:299-309)."""
    for module in model.modules():
        if type(module) in targ_type:
            module.set_update_stat(update_stat)
    return model


def update_quant_range(model, data, graph, bottoms, is_detection=False):
    """Run the distilled batches through the model so every QuantMeasure records its range
    (improve_dfq.py:280-297); the first layer's range is pinned to the ImageNet-normalised image
    range (2.64 / -2.11790393), or +-1 for detection."""
    with torch.no_grad():
        for batch in data:
            dev = next(model.parameters()).device
            model(batch.to(dev))
    for key in graph:
        bot = bottoms[key]
        if bot is not None and bot[0] == 'Data' and hasattr(graph[key], 'quant'):
            q = graph[key].quant
            if is_detection:
                q.running_max.fill_(1.0)
                q.running_min.fill_(-1.0)
            else:
                q.running_max.fill_(2.64)
                q.running_min.fill_(-2.11790393)
    return model
