"""The int8 calibration table the reference hands to ncnn2int8 (convert_ncnn.py:178-201).

Two blocks of lines, both in graph order over the conv / linear layers:
  * weights:     ``<name>_param_0 s s ... s``  with s = 128 / max(|min W|, |max W|), written once per output
                 channel (the reference derives ONE scale per layer and repeats it);
  * activations: ``<name> s``  with s = 128 / max(|running_min|, |running_max|) of the layer's input quantiser
                 (filled by set_quant_minmax or update_quant_range).
The reference takes the line names from the table ncnn2table wrote for the same .param file; callers that have
that table pass its first column as ``names`` (weights block first, then activations), otherwise the graph keys
are used.  Weight ranges of the whole network come from one multi-tensor min/max launch (dfq_quant_plan_measure).
``per_channel=True`` (extension) writes a genuine scale per output channel from
dfq_row_range's |max| per row.
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn

from . import _ffi
from . import prims
from .dfq import _RawDeviceBuffer


def weight_ranges(graph, targ_type=(nn.Conv2d, nn.Linear)):
    """{key: (min, max)} of every targ layer's weight: one launch + one read-back for the whole network."""
    lib = _ffi.lib()
    keys = [k for k in graph if type(graph[k]) in tuple(targ_type)]
    if not keys:
        return {}
    with torch.no_grad():
        stage = _ffi.Stage()
        keep = [stage.bind(graph[k].weight) for k in keys]
        arr = (_ffi.DfqSegment * len(keys))(*[_ffi.DfqSegment(w.data_ptr(), w.numel(), 8, 0, None) for w in keep])
        plan = ctypes.c_void_p()
        _ffi.check(lib.dfq_quant_plan_create(arr, len(keys), ctypes.byref(plan)))
        try:
            _ffi.check(lib.dfq_quant_plan_measure(plan, _ffi.stream_arg()))
            _ffi.synchronize()
            addr = lib.dfq_quant_plan_minmax(plan)
            vals = _RawDeviceBuffer(addr, 2 * len(keys), stage.device).tensor().tolist()
        finally:
            lib.dfq_quant_plan_destroy(plan)
    return {k: (vals[2 * i], vals[2 * i + 1]) for i, k in enumerate(keys)}


def calibration_table(graph, targ_type=(nn.Conv2d, nn.Linear), names=None, per_channel=False):
    """Lines of ``model_int8_tensor.table`` (convert_ncnn.py:180-201)."""
    keys = [k for k in graph if type(graph[k]) in tuple(targ_type)]
    ranges = weight_ranges(graph, targ_type)
    if names is None:
        names = ['{}_param_0'.format(k) for k in keys] + [str(k) for k in keys]
    assert len(names) == 2 * len(keys), 'need one name per layer for the weight block and one for the activation block'
    lines = []
    for i, k in enumerate(keys):
        layer = graph[k]
        if per_channel:
            amax = prims.row_range(layer.weight, signed=True).tolist()
            lines.append(' '.join([names[i]] + [str(128. / a) for a in amax]))
        else:
            mi, ma = ranges[k]
            scale = 128. / (max(abs(ma), abs(mi)))
            lines.append(' '.join([names[i]] + [str(scale)] * layer.weight.shape[0]))
    for i, k in enumerate(keys):
        q = graph[k].quant
        mi = float(torch.min(q.running_min))
        ma = float(torch.max(q.running_max))
        scale = 128. / (max(abs(ma), abs(mi)))
        lines.append(' '.join([names[len(keys) + i], str(scale)]))
    return lines


def write_calibration_table(path, graph, targ_type=(nn.Conv2d, nn.Linear), names=None, per_channel=False):
    lines = calibration_table(graph, targ_type, names, per_channel)
    with open(path, 'w') as f:
        for line in lines:
            f.write(line + '\n')
    return lines
