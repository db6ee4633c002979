#!/usr/bin/env python3
"""The calibration section of the reference's main_cls.py (lines 116-188) with dfq_amd as the engine.

    python examples/calibrate.py [--net mobilenet_v2] [--seed 0] [--bits-weight 8] [--bits-bias 16]
                                 [--absorption] [--distill-range] [--table model_int8_tensor.table]

A synthetic, randomly initialised network stands in for the pretrained checkpoint (no network access);
everything after "model built" is what a user of the reference runs, with only the imports changed
(INTEGRATION.md): trace -> fold BN -> pair layers -> cross-layer equalisation -> [bias absorption] ->
bias correction -> weight/bias fake-quant -> analytic activation ranges -> ncnn calibration table; with
--distill-range (main_cls.py:86-113, :183-186): ZeroQ-distilled batches from the UNFOLDED model's BatchNorm statistics, then
activation ranges recorded by running them through the quantised model instead of the analytic ranges.
Needs an MI355X (the engine has no CPU path).
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from dfq_amd import ncnn_table, synthetic                                              # noqa: E402
from dfq_amd.dfq import bias_absorption, bias_correction, cross_layer_equalization    # noqa: E402
from dfq_amd.improve_dfq import _swap_modules, set_update_stat, update_quant_range      # noqa: E402
from dfq_amd.utils.quantize import QuantMeasure                                        # noqa: E402
from dfq_amd.zeroq import getDistilData                                                # noqa: E402
from dfq_amd.utils.layer_transform import merge_batchnorm, quantize_targ_layer, set_quant_minmax   # noqa: E402
from dfq_amd.utils.quantize import QConv2d, QLinear                                    # noqa: E402
from dfq_amd.utils.relation import create_relation                                     # noqa: E402
import dfq_amd.dfq as engine                                                           # noqa: E402


def switch_layers(model, graph):
    """Conv2d -> QConv2d, Linear -> QLinear, weights shared (what the reference's switch_layers does through
    PyTransformer, main_cls.py:116-117), keeping `graph` pointing at the new modules."""
    mapping = {}
    for m in model.modules():
        if type(m) == nn.Conv2d:
            q = QConv2d(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, m.dilation, m.groups,
                        m.bias is not None)
        elif type(m) == nn.Linear:
            q = QLinear(m.in_features, m.out_features, m.bias is not None)
        else:
            continue
        q.weight = m.weight
        q.bias = m.bias
        mapping[m] = q.to(m.weight.device)
    _swap_modules(model, mapping)
    for k in graph:
        if not isinstance(graph[k], str) and graph[k] in mapping:
            graph[k] = mapping[graph[k]]
    return model


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--net', default='mobilenet_v2')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--bits-weight', type=int, default=8)
    ap.add_argument('--bits-bias', type=int, default=16)
    ap.add_argument('--absorption', action='store_true')
    ap.add_argument('--max-sweeps', type=int, default=None)
    ap.add_argument('--distill-range', action='store_true', help='activation ranges from ZeroQ-distilled batches (config 5)')
    ap.add_argument('--dis-batch-size', type=int, default=8)
    ap.add_argument('--dis-num-batch', type=int, default=2)
    ap.add_argument('--dis-iterations', type=int, default=20, help='the reference runs up to 1000 per batch')
    ap.add_argument('--image-size', type=int, default=32)
    ap.add_argument('--table', default=None, help='write the ncnn int8 calibration table here')
    ap.add_argument('--device', default='cuda')
    args = ap.parse_args(argv)

    model, graph, bottoms = synthetic.build(args.net, seed=args.seed)      # main_cls.py:91-135 (model + traced graph)
    model.to(args.device)
    data_distill = None
    if args.distill_range:                                                                           # :86-100
        import copy
        model_original = copy.deepcopy(model)          # BatchNorm still unfolded: its running statistics are the target
        data_distill = getDistilData(model_original, (args.dis_batch_size, 3, args.image_size, args.image_size),
                                     num_batch=args.dis_num_batch, bn_merged=False, iterations=args.dis_iterations,
                                     generator=torch.Generator().manual_seed(args.seed))
    switch_layers(model, graph)
    targ_layer = [QConv2d, QLinear]

    t0 = time.perf_counter()
    model = merge_batchnorm(model, graph, bottoms, targ_layer)                                      # :149
    res = create_relation(graph, bottoms, targ_layer, delete_single=False)                          # :152
    cross_layer_equalization(graph, res, targ_layer, visualize_state=False, converge_thres=2e-7,
                             max_sweeps=args.max_sweeps)                                            # :153
    sweeps = engine.last_equalization['sweeps']
    if args.absorption:
        bias_absorption(graph, res, bottoms, 3)                                                     # :156
    bias_correction(graph, bottoms, targ_layer, bits_weight=args.bits_weight)                       # :175
    if args.distill_range:                                                                           # :183-186
        set_update_stat(model, [QuantMeasure], True)
        model = update_quant_range(model, data_distill, graph, bottoms)
        set_update_stat(model, [QuantMeasure], False)
    else:
        graph = quantize_targ_layer(graph, args.bits_weight, args.bits_bias, targ_layer)            # :181
        set_quant_minmax(graph, bottoms, verbose=False)                                             # :188
    if args.device == 'cuda':
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_w = sum(graph[k].weight.numel() for k in graph if type(graph[k]) in targ_layer)
    levels = max(len(torch.unique(graph[k].weight)) for k in graph if type(graph[k]) in targ_layer) if not args.distill_range else -1
    print('{}: {} layers, {} weights, {} relations; {} equalisation sweeps; <= {} distinct weight levels per layer; '
          '{:.1f} ms wall for the whole calibration section'.format(args.net, sum(type(graph[k]) in targ_layer for k in graph),
                                                                     n_w, len(res), sweeps, levels, dt * 1e3))
    if args.table:
        lines = ncnn_table.write_calibration_table(args.table, graph, targ_type=targ_layer)
        print('wrote {} lines to {}'.format(len(lines), args.table))
    return model, graph, bottoms


if __name__ == '__main__':
    main()
