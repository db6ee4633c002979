#!/usr/bin/env python
"""SHA-256 over every tensor a bias-correction pass leaves behind (biases, BN proxies, correction vectors), per network:
   two builds of the library that print the same digests perform the same arithmetic bit for bit.
   DFQ_HIP_LIB=variants/libdfq_hip_x.so python tools/bc_digest.py mobilenet_v2 resnet18 deeplab_mnv2"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

dev = torch.device('cuda', 0) if torch.cuda.is_available() else torch.device('cpu')
for net in sys.argv[1:] or ['mobilenet_v2']:
    proto = bench.prepare(net, 0, dev)
    unit = bench.make_unit([proto])
    unit['le'].enqueue(3, restart=True, max_sweeps=3, converge_thres=-1.0, converge_count=10 ** 9)
    unit['le'].query()
    for signed in (False, True):
        unit['bc'].run(signed=signed)
        unit['bc'].status()
        h = hashlib.sha256()
        for _, graph, bottoms, _ in unit['nets']:
            for k in sorted(graph, key=str):
                m = graph[k]
                for name in ('weight', 'bias', 'fake_weight', 'fake_bias'):
                    t = getattr(m, name, None)
                    if isinstance(t, torch.Tensor):
                        h.update(t.detach().cpu().contiguous().numpy().tobytes())
        print(net, 'signed' if signed else 'unsigned', h.hexdigest()[:32])
