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


def update_quant_range(model, data, graph, bottoms, is_detection=False, group=None):
    """Run the distilled batches through the model so every QuantMeasure records its range
    (improve_dfq.py:280-297); the first layer's range is pinned to the ImageNet-normalised image
    range (2.64 / -2.11790393), or +-1 for detection.

    ``group`` (extension; SURVEY 8e "C5: data-parallel over distilled batches"): a torch.distributed process group whose ranks
    each hold the same model.  Rank r then runs batches r, r + world, ... only, and ONE all_reduce merges the [modules, 2] table
    of running ranges (max of the maxima, min of the minima) at the end, so every rank ends with the same ranges.  Not
    bit-identical to the sequential pass, and it cannot be: the reference quantises every batch with the range recorded SO FAR
    (quantize.py:103-119), so what a later layer sees of batch k depends on the batches in front of it; a rank that has seen
    fewer batches quantises with a slightly narrower range.  The effect is second order -- half a quantisation step of the
    producing layer, 1/510 of its range -- tests/test_range_parity.py holds the merged ranges within 2 % of the sequential ones."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if group is not None else 1
    rank = dist.get_rank(group) if group is not None else 0
    with torch.no_grad():
        for i, batch in enumerate(data):
            if i % world != rank:
                continue
            dev = next(model.parameters()).device
            model(batch.to(dev))
        if world > 1:
            measures = [m for m in model.modules() if isinstance(m, QuantMeasure)]
            if measures:
                dev = measures[0].running_max.device
                table = torch.stack([torch.cat([m.running_max.reshape(1).to(dev), -m.running_min.reshape(1).to(dev)])
                                     for m in measures])                               # max and -min: ONE reduction (MAX)
                comm = table if dist.get_backend(group) == 'nccl' else table.cpu()
                dist.all_reduce(comm, op=dist.ReduceOp.MAX, group=group)
                table = comm.to(dev)
                for m, row in zip(measures, table):
                    m.running_max.copy_(row[0:1].to(m.running_max.device))
                    m.running_min.copy_((-row[1:2]).to(m.running_min.device))
    for key in graph:
        bot = bottoms[key]
        # (the graph fxgraph.quantize_tensor_ops returns lists a layer's input quantiser as a node of its own: there the node fed
        # by 'Data' IS the QuantMeasure of the first layer)
        first = graph[key].quant if hasattr(graph[key], 'quant') else (graph[key] if isinstance(graph[key], QuantMeasure) else None)
        if bot is not None and bot[0] == 'Data' and first is not None:
            q = first
            if is_detection:
                q.running_max.fill_(1.0)
                q.running_min.fill_(-1.0)
            else:
                q.running_max.fill_(2.64)
                q.running_min.fill_(-2.11790393)
    return model
