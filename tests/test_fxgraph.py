"""The torch.fx tracer that stands in for the reference's absent PyTransformer (row f4, dfq_amd/fxgraph.py): graph format,
and -- ADVICE round 1 -- that functional activations and unknown operations are NOT silently dropped."""
import pytest
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


# ----------------------------------------------------------------------------------------------------------------------
# Pins against the only artefacts the reference holds for its (absent) tracer: the rendered graphs
# /root/reference/images/graph_cls.png and images/graph_deeplab.png (node name = '<kind>_<index>', 'Bottoms: ...').
# The entries below are transcribed from those images.  graph_cls.png was rendered after switch_layers
# (QuantConv2d / QuantLinear class names; --relu) and graph_deeplab.png without --relu (ReLU6 in the backbone).
# ----------------------------------------------------------------------------------------------------------------------
_CLS_TAIL = {            # images/graph_cls.png, last 16 nodes (Quant* = the swapped classes of Conv2d / Linear)
    'Conv2d_136': ['ReLU_135'], 'BatchNorm2d_137': ['Conv2d_136'], 'add_138': ['add_129', 'BatchNorm2d_137'],
    'Conv2d_139': ['add_138'], 'BatchNorm2d_140': ['Conv2d_139'], 'ReLU_141': ['BatchNorm2d_140'],
    'Conv2d_142': ['ReLU_141'], 'BatchNorm2d_143': ['Conv2d_142'], 'ReLU_144': ['BatchNorm2d_143'],
    'Conv2d_145': ['ReLU_144'], 'BatchNorm2d_146': ['Conv2d_145'], 'Conv2d_147': ['BatchNorm2d_146'],
    'BatchNorm2d_148': ['Conv2d_147'], 'ReLU_149': ['BatchNorm2d_148'], 'torch.mean_150': ['ReLU_149'],
    'Linear_151': ['torch.mean_150'],
}

_DEEPLAB_HEAD = {        # images/graph_deeplab.png, first 20 nodes
    'Conv2d_1': ['Data'], 'BatchNorm2d_2': ['Conv2d_1'], 'ReLU6_3': ['BatchNorm2d_2'], 'F.pad_4': ['ReLU6_3'],
    'Conv2d_5': ['F.pad_4'], 'BatchNorm2d_6': ['Conv2d_5'], 'ReLU6_7': ['BatchNorm2d_6'], 'Conv2d_8': ['ReLU6_7'],
    'BatchNorm2d_9': ['Conv2d_8'], 'F.pad_10': ['BatchNorm2d_9'], 'Conv2d_11': ['F.pad_10'],
    'BatchNorm2d_12': ['Conv2d_11'], 'ReLU6_13': ['BatchNorm2d_12'], 'Conv2d_14': ['ReLU6_13'],
    'BatchNorm2d_15': ['Conv2d_14'], 'ReLU6_16': ['BatchNorm2d_15'], 'Conv2d_17': ['ReLU6_16'],
    'BatchNorm2d_18': ['Conv2d_17'], 'F.pad_19': ['BatchNorm2d_18'],
}
_DEEPLAB_MID = {         # residual joins of the last backbone stage
    'add_144': ['BatchNorm2d_134', 'BatchNorm2d_143'], 'F.pad_145': ['add_144'], 'Conv2d_146': ['F.pad_145'],
    'add_154': ['add_144', 'BatchNorm2d_153'], 'F.pad_155': ['add_154'], 'Conv2d_162': ['ReLU6_161'],
}
_DEEPLAB_TAIL = {        # ASPP + decoder: every node from BatchNorm2d_163 to the end
    'BatchNorm2d_163': ['Conv2d_162'],
    'Conv2d_164': ['BatchNorm2d_163'], 'BatchNorm2d_165': ['Conv2d_164'], 'ReLU_166': ['BatchNorm2d_165'],
    'Conv2d_167': ['BatchNorm2d_163'], 'BatchNorm2d_168': ['Conv2d_167'], 'ReLU_169': ['BatchNorm2d_168'],
    'Conv2d_170': ['BatchNorm2d_163'], 'BatchNorm2d_171': ['Conv2d_170'], 'ReLU_172': ['BatchNorm2d_171'],
    'Conv2d_173': ['BatchNorm2d_163'], 'BatchNorm2d_174': ['Conv2d_173'], 'ReLU_175': ['BatchNorm2d_174'],
    'AdaptiveAvgPool2d_176': ['BatchNorm2d_163'], 'Conv2d_177': ['AdaptiveAvgPool2d_176'],
    'BatchNorm2d_178': ['Conv2d_177'], 'ReLU_179': ['BatchNorm2d_178'], 'F.interpolate_180': ['ReLU_179'],
    'torch.cat_181': ['ReLU_166', 'ReLU_169', 'ReLU_172', 'ReLU_175', 'F.interpolate_180'],
    'Conv2d_182': ['torch.cat_181'], 'BatchNorm2d_183': ['Conv2d_182'], 'ReLU_184': ['BatchNorm2d_183'],
    'Dropout_185': ['ReLU_184'],
    'F.interpolate_189': ['Dropout_185'], 'torch.cat_190': ['F.interpolate_189', 'ReLU_188'],
    'Conv2d_191': ['torch.cat_190'], 'BatchNorm2d_192': ['Conv2d_191'], 'ReLU_193': ['BatchNorm2d_192'],
    'Dropout_194': ['ReLU_193'], 'Conv2d_195': ['Dropout_194'], 'BatchNorm2d_196': ['Conv2d_195'],
    'ReLU_197': ['BatchNorm2d_196'], 'Dropout_198': ['ReLU_197'], 'Conv2d_199': ['Dropout_198'],
    'F.interpolate_200': ['Conv2d_199'],
}


def _check_pins(graph, bottoms, pins):
    for key, bots in pins.items():
        assert key in graph, key
        assert bottoms[key] == bots, (key, bottoms[key], bots)
        kind = key.rsplit('_', 1)[0]
        if isinstance(graph[key], str):
            assert graph[key] == key
        else:
            assert type(graph[key]).__name__ == kind


def test_mobilenet_v2_graph_is_the_references():
    """images/graph_cls.png: 152 entries incl. Data, the last one (Quant)Linear_151 behind torch.mean_150."""
    from dfq_amd import synthetic
    _, graph, bottoms = synthetic.build('mobilenet_v2', seed=0)
    keys = list(graph)
    assert len(keys) == 152 and keys[0] == 'Data' and keys[-1] == 'Linear_151'
    assert [int(k.rsplit('_', 1)[1]) for k in keys[1:]] == list(range(1, 152))
    _check_pins(graph, bottoms, _CLS_TAIL)
    assert len(rel.create_relation(graph, bottoms, TARG)) == 37


def test_deeplab_graph_is_the_references():
    """images/graph_deeplab.png: 201 entries; Dropout modules ARE nodes (185, 194, 198), which is why
    F.interpolate / torch.cat carry the numbers 180 / 181 / 189 / 190 / 200 and why relation.py:36-46 never pairs the
    decoder's 3x3 convs: 35 relations, all in the backbone."""
    from dfq_amd import synthetic
    _, graph, bottoms = synthetic.build('deeplab_mnv2', seed=0, keep_relu6=True)
    keys = list(graph)
    assert len(keys) == 201 and keys[0] == 'Data' and keys[-1] == 'F.interpolate_200'
    assert [int(k.rsplit('_', 1)[1]) for k in keys[1:]] == list(range(1, 201))
    for pins in (_DEEPLAB_HEAD, _DEEPLAB_MID, _DEEPLAB_TAIL):
        _check_pins(graph, bottoms, pins)
    assert [k for k in keys if isinstance(graph[k], str) and ('interpolate' in k or 'cat' in k)] == \
        ['F.interpolate_180', 'torch.cat_181', 'F.interpolate_189', 'torch.cat_190', 'F.interpolate_200']
    assert [k for k in keys if isinstance(graph[k], nn.Dropout)] == ['Dropout_185', 'Dropout_194', 'Dropout_198']

    # --relu (what the benchmark configuration runs): same topology, 35 relations, none touching the decoder
    _, graph, bottoms = synthetic.build('deeplab_mnv2', seed=0)
    rels = rel.create_relation(graph, bottoms, TARG)
    assert len(rels) == 35
    touched = {k for r in rels for k in r.get_idxs()[:2]}
    assert not touched & {'Conv2d_182', 'Conv2d_186', 'Conv2d_191', 'Conv2d_195', 'Conv2d_199'}
    assert max(int(k.rsplit('_', 1)[1]) for k in touched) <= 162


def test_dropout_module_stops_the_pairing_walk():
    """relation.py:36-46: the walk from a layer to its predecessor passes BN / ReLU / QuantMeasure / AvgPool2d /
    F.pad / torch.mean only -- a Dropout module between two convs means no relation, although it is the identity in eval."""
    net = nn.Sequential(nn.Conv2d(3, 4, 1), nn.BatchNorm2d(4), nn.ReLU(), nn.Dropout(0.5), nn.Conv2d(4, 4, 1),
                        nn.BatchNorm2d(4), nn.ReLU(), nn.Conv2d(4, 2, 1)).eval()
    graph, bottoms = fxgraph.trace(net)
    assert [type(v).__name__ for v in graph.values() if not isinstance(v, str)].count('Dropout') == 1
    pairs = [(r.get_idxs()[0], r.get_idxs()[1]) for r in rel.create_relation(graph, bottoms, TARG)]
    assert pairs == [('Conv2d_5', 'Conv2d_8')]


@pytest.mark.parametrize('name', ['mobilenet_v2', 'deeplab_mnv2'])
@pytest.mark.parametrize('relu', [False, True])
def test_synthetic_graph_equals_the_reference_model_class(name, relu):
    """tests/golden/graph_*.json (oracle/make_golden_graphs.py): the REFERENCE's model definitions traced with this
    tracer -- node keys, bottoms, conv / linear / BN geometry -- and the relation triples the UNMODIFIED
    utils/relation.py:create_relation finds on them.  The synthetic networks the bench and the full-size fixtures
    calibrate must be that graph, and this repo's create_relation must find those relations."""
    import json
    import os
    from dfq_amd import synthetic
    from tests.common import GOLD
    with open(os.path.join(GOLD, 'graph_{}{}.json'.format(name, '_relu' if relu else ''))) as f:
        ref = json.load(f)
    _, graph, bottoms = synthetic.build(name, seed=0, keep_relu6=not relu)
    assert list(graph) == [n['key'] for n in ref['nodes']]
    for n in ref['nodes']:
        k, m = n['key'], graph[n['key']]
        assert bottoms[k] == n['bottoms'], k
        if n['kind'] == 'str':
            assert m == k
            continue
        assert type(m).__name__ == n['kind'], k
        if isinstance(m, nn.Conv2d):
            assert [list(m.weight.shape), m.groups, list(m.stride), list(m.padding), list(m.dilation),
                    m.bias is not None] == n['geom'], k
        elif isinstance(m, nn.Linear):
            assert [list(m.weight.shape), m.bias is not None] == n['geom'], k
        elif isinstance(m, nn.BatchNorm2d):
            assert [m.num_features] == n['geom'], k
    rels = rel.create_relation(graph, bottoms, TARG)
    assert [list(r.get_idxs()) for r in rels] == ref['relations']
    assert len(rels) == {('mobilenet_v2', True): 37, ('deeplab_mnv2', True): 35,
                         ('mobilenet_v2', False): 2, ('deeplab_mnv2', False): 1}[(name, relu)]
