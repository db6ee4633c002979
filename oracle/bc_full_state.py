"""TEST INFRASTRUCTURE: the common input state of the full-size, stage-wise bias-correction check (see
oracle/make_golden_bc_full.py): synthetic network (seed 0) -> BN folded and equalised by the numpy oracle -> the same
numbers written into the torch modules of the graph.  Deterministic numpy / torch-CPU arithmetic: the generator (build
container) and the tests (CPU, GPU box) reproduce it; `input_moments` fingerprints it."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from dfq_amd import synthetic
from oracle import dfq_oracle as orc
from oracle import graphspec

TARG = [nn.Conv2d, nn.Linear]


def build(name, sweeps=None):
    model, graph, bottoms = synthetic.build(name, seed=0)
    spec = graphspec.from_torch(graph, bottoms, TARG)
    orc.merge_batchnorm(spec)
    rels = orc.create_relation(spec)
    if sweeps:
        orc.cross_layer_equalization(spec, rels, max_sweeps=sweeps, converge_thres=-1.0, converge_count=10 ** 9)
    else:
        orc.cross_layer_equalization(spec, rels)
    with torch.no_grad():
        for k in graph:
            n = spec.nodes[k]
            if n.kind == 'targ':
                graph[k].weight.copy_(torch.from_numpy(n.weight))
                if n.bias is not None:
                    if graph[k].bias is None:
                        graph[k].bias = nn.Parameter(torch.zeros(n.bias.shape[0]), requires_grad=False)
                    graph[k].bias.copy_(torch.from_numpy(n.bias))
            elif n.kind == 'bn' and n.fake_weight is not None:
                graph[k].register_buffer('fake_weight', torch.from_numpy(n.fake_weight.copy()))
                graph[k].register_buffer('fake_bias', torch.from_numpy(n.fake_bias.copy()))
    return model, graph, bottoms, spec


def input_moments(spec):
    """{'in.L<i>.<w|b|fw|fb>': float64 [min, max, sum, sum of squares]} of every tensor bias correction reads."""
    out = {}
    for i, k in enumerate(spec.order):
        n = spec.nodes[k]
        items = []
        if n.kind == 'targ':
            items = [('w', n.weight), ('b', n.bias)]
        elif n.kind == 'bn' and n.fake_weight is not None:
            items = [('fw', n.fake_weight), ('fb', n.fake_bias)]
        for tag, v in items:
            if v is not None:
                x = np.asarray(v, dtype=np.float64)
                out['in.L{}.{}'.format(i, tag)] = np.array([x.min(), x.max(), x.sum(), (x * x).sum()])
    return out
