#!/usr/bin/env python
"""config 5 end to end (bench.distill_range_pass) alone: python tools/distill_probe.py [net:batches:shape]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

spec = sys.argv[1] if len(sys.argv) > 1 else 'mobilenet_v2:8:64,3,224,224'
net, n, shape = spec.split(':')
rec = bench.distill_range_pass(net, [int(v) for v in shape.split(',')], int(n), torch.device('cuda', 0))
print(json.dumps({k: rec[k] for k in ('ms_per_batch', 'convolutions_only_ms_per_batch', 'quant_measure_ms_per_batch',
                                       'quant_measure_GBps', 'quant_measure_frac_of_hbm_peak', 'wall_ms_total', 'ms_total')}))
