"""The torch.fx tracer that stands in for the reference's absent PyTransformer (row f4, dfq_amd/fxgraph.py): graph format,
and -- ADVICE round 1 -- that functional activations and unknown operations are NOT silently dropped."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from dfq_amd import fxgraph
from dfq_amd.utils import relation as rel

TARG = [nn.Conv2d, nn.Linear]


class FunctionalNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.c0 = nn.Conv2d(3, 8, 3, padding=1)
        self.b0 = nn.BatchNorm2d(8)
        self.c1 = nn.Conv2d(8, 8, 1)
        self.b1 = nn.BatchNorm2d(8)
        self.c2 = nn.Conv2d(8, 8, 1)
        self.b2 = nn.BatchNorm2d(8)
        self.c3 = nn.Conv2d(8, 8, 1)
        self.b3 = nn.BatchNorm2d(8)
        self.c4 = nn.Conv2d(8, 4, 1)
        self.fc = nn.Linear(4, 2)

    def forward(self, x):
        x = F.relu(self.b0(self.c0(x)))                 # ReLU: scale-commuting, relation c0 -> c1
        x = F.relu6(self.b1(self.c1(x)))                # ReLU6: NOT passed through by relation.py:36-41
        x = self.b2(self.c2(x))
        x = x * torch.sigmoid(x)                        # unknown ops: opaque nodes
        x = self.b3(self.c3(x)).relu()                  # Tensor.relu -> ReLU
        x = F.dropout(self.c4(x), 0.5, self.training)   # eval: transparent
        return self.fc(torch.flatten(x.mean((2, 3)), 1))


def test_functional_ops_are_kept():
    net = FunctionalNet().eval()
    graph, bottoms = fxgraph.trace(net)
    kinds = [type(v).__name__ if not isinstance(v, str) else v.split('_')[0] for v in graph.values()]
    assert kinds.count('ReLU') == 2 and kinds.count('ReLU6') == 1
    # unknown ops are opaque string nodes; their KEYS never carry the operation's name (the passes classify string nodes
    # by substrings of the key: an `addmm` or `scatter` must not read as a residual add / a concat), the values do
    opaque = {k: v for k, v in graph.items() if isinstance(v, str) and k.startswith('opaque_')}
    assert sorted(v.split(':')[1] for v in opaque.values()) == ['mul', 'sigmoid']
    assert not any(tag in k for k in opaque for tag in ('add', 'cat', 'mean', 'pad', 'interpolate', 'softmax'))
    assert not any('dropout' in str(k) or 'flatten' in str(k) for k in graph)
    key = {m: k for k, m in graph.items() if not isinstance(m, str)}
    # the ReLU sits between b0 and c1; the ReLU6 between b1 and c2
    assert type(graph[bottoms[key[net.c1]][0]]) is nn.ReLU and bottoms[bottoms[key[net.c1]][0]] == [key[net.b0]]
    assert type(graph[bottoms[key[net.c2]][0]]) is nn.ReLU6
    # the linear layer sees the mean node (flatten is transparent), the mean sees c4 (dropout is transparent)
    mean_key = bottoms[key[net.fc]][0]
    assert mean_key.startswith('torch.mean') and bottoms[mean_key] == [key[net.c4]]

    rels = rel.create_relation(graph, bottoms, TARG)
    pairs = {(r.get_idxs()[0], r.get_idxs()[1]) for r in rels}
    assert (key[net.c0], key[net.c1]) in pairs                     # across F.relu
    assert (key[net.c1], key[net.c2]) not in pairs                 # never across ReLU6
    assert (key[net.c2], key[net.c3]) not in pairs                 # never across sigmoid / mul
    assert (key[net.c3], key[net.c4]) in pairs                     # across Tensor.relu
    assert (key[net.c4], key[net.fc]) in pairs                     # across torch.mean (relation.py:40)


def test_relu_is_seen_by_bias_absorption_walk():
    from dfq_amd.dfq import _relu_between
    net = FunctionalNet().eval()
    graph, bottoms = fxgraph.trace(net)
    key = {m: k for k, m in graph.items() if not isinstance(m, str)}
    assert _relu_between(graph, bottoms, key[net.c1], key[net.c0])
    assert not _relu_between(graph, bottoms, key[net.fc], key[net.c4])
