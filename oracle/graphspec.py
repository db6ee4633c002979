"""Torch-free graph container used by the CPU oracle (TEST INFRASTRUCTURE ONLY).

The reference keeps the traced network as two ``OrderedDict``s (``graph``: key -> nn.Module | str,
``bottoms``: key -> list[key] | None; main_cls.py:134-135).  ``GraphSpec`` holds the same
topology with numpy arrays instead of modules so the oracle needs no torch.

Node kinds: 'data', 'targ' (Conv2d/Linear family), 'bn', 'relu', 'relu6', 'qm' (QuantMeasure), 'avgpool',
'op' (string-valued tensor op such as 'add_12', 'torch.cat_30', 'F.pad_7', 'torch.mean_150'),
'other'.
"""
from __future__ import annotations

import copy
from collections import OrderedDict

import numpy as np

F32 = np.float32


class Node:
    __slots__ = ('key', 'kind', 'weight', 'bias', 'groups', 'gamma', 'beta', 'mean', 'var', 'eps',
                 'fake_weight', 'fake_bias', 'linear')

    def __init__(self, key, kind):
        self.key = key
        self.kind = kind
        self.weight = None
        self.bias = None
        self.groups = 1
        self.linear = False
        self.gamma = self.beta = self.mean = self.var = None
        self.eps = 0.0
        self.fake_weight = None
        self.fake_bias = None


class GraphSpec:
    def __init__(self):
        self.order = []                 # keys in graph (forward) order
        self.nodes = OrderedDict()      # key -> Node
        self.bottoms = OrderedDict()    # key -> list[key] | None

    def add(self, node, bottoms):
        self.order.append(node.key)
        self.nodes[node.key] = node
        self.bottoms[node.key] = None if bottoms is None else list(bottoms)
        return node

    def clone(self):
        return copy.deepcopy(self)

    def targ_keys(self):
        return [k for k in self.order if self.nodes[k].kind == 'targ']

    def n_weights(self):
        return int(sum(self.nodes[k].weight.size for k in self.targ_keys()))


def _np(t):
    return None if t is None else np.ascontiguousarray(t.detach().cpu().numpy().astype(F32)).copy()


def from_torch(graph, bottoms, targ_type):
    """Convert the reference-format (graph, bottoms) dicts (torch modules) into a GraphSpec.

    Module keys are kept as they are (any hashable).  ``targ_type`` is the list of layer classes
    the reference calls ``targ_layer`` (main_cls.py:137-145).
    """
    import torch.nn as nn
    spec = GraphSpec()
    for key in graph:
        m = graph[key]
        bots = bottoms[key]
        if isinstance(m, str):
            kind = 'data' if key == 'Data' else 'op'
            node = Node(key, kind)
        elif type(m) in targ_type:
            node = Node(key, 'targ')
            node.weight = _np(m.weight)
            node.bias = _np(m.bias) if m.bias is not None else None
            node.groups = getattr(m, 'groups', 1)
            node.linear = isinstance(m, nn.Linear)
        elif type(m) == nn.BatchNorm2d:
            node = Node(key, 'bn')
            node.gamma = _np(m.weight)
            node.beta = _np(m.bias)
            node.mean = _np(m.running_mean)
            node.var = _np(m.running_var)
            node.eps = float(m.eps)
            if hasattr(m, 'fake_weight'):
                node.fake_weight = _np(m.fake_weight)
                node.fake_bias = _np(m.fake_bias)
        elif type(m) == nn.ReLU:
            node = Node(key, 'relu')
        elif type(m) == nn.ReLU6:
            node = Node(key, 'relu6')
        elif type(m) == nn.AvgPool2d:
            node = Node(key, 'avgpool')
        elif type(m).__name__ == 'QuantMeasure':
            node = Node(key, 'qm')
        else:
            node = Node(key, 'other')
        spec.add(node, bots)
    return spec
