"""The drop-in claim of INTEGRATION.md section 1 ("only the imports change"), with the REFERENCE'S OWN OBJECTS.

The unmodified reference (oracle/_ref: its five hot-path modules byte-compiled from /root/reference by
oracle/build_ref.py -- present in the build container and, as a travelling build artefact, on the GPU box) prepares the
graph the way main_cls.py:116-149 does: its `QuantNConv2d` / `QuantNLinear` (or plain `nn.Conv2d` / `nn.Linear`) modules,
ITS `merge_batchnorm` (utils/layer_transform.py:231-276) and ITS `create_relation` (utils/relation.py:30-94) -- so the
`Relation` instances are utils/relation.py:5-27's, not this package's.  That graph and those relations are then handed

  * to the reference's own `cross_layer_equalization` + `bias_correction` (dfq.py:78-117, 173-293), and
  * a deep copy of them to `dfq_amd.dfq.cross_layer_equalization` + `bias_correction` (main_cls.py:149-181 with the
    imports changed),

and every weight, bias, BN proxy and the `S` the relations accumulated must agree within 1e-5 (the float32 contract of
BASELINE.json; LE cannot be bit-exact against torch's CPU sqrt), with the same sweep count.  Bias correction
is compared from the reference's post-LE state (stage-wise).

Skipped when oracle/_ref is absent (a checkout that never ran __graft_entry__.build() next to /root/reference).
"""
import copy
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

from common import assert_close, npy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, 'oracle', '_ref')

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REFDIR, 'dfq.pyc')),
                                reason='oracle/_ref is not built (oracle/build_ref.py needs /root/reference)')


@pytest.fixture
def ref():
    """The reference's modules, imported from oracle/_ref for the duration of one test (its top-level `dfq` and `utils`
    packages are taken out of sys.modules again afterwards)."""
    before = set(sys.modules)
    sys.path.insert(0, REFDIR)
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    try:
        import dfq as ref_dfq
        from utils import layer_transform as ref_lt
        from utils import quantize as ref_q
        from utils import relation as ref_rel
        assert os.path.dirname(os.path.abspath(ref_dfq.__file__)) == REFDIR, ref_dfq.__file__

        class Ref:
            dfq, lt, q, rel = ref_dfq, ref_lt, ref_q, ref_rel
        yield Ref
    finally:
        sys.dont_write_bytecode = old
        sys.path.remove(REFDIR)
        for name in set(sys.modules) - before:
            if name == 'dfq' or name == 'utils' or name.startswith('utils.'):
                del sys.modules[name]


def _swap_to_reference_layers(model, graph, ref_q):
    """What the reference's switch_layers (utils/layer_transform.py:151-188, PyTransformer's trans_layers) leaves behind for
    `--quantize`: every nn.Conv2d / nn.Linear replaced by the reference's QuantNConv2d / QuantNLinear holding the same
    parameters, in the model and in the graph."""
    swapped = {}
    for name, m in list(model.named_modules()):
        if type(m) == nn.Conv2d:
            q = ref_q.QuantNConv2d(m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, m.dilation, m.groups,
                                   m.bias is not None)
        elif type(m) == nn.Linear:
            q = ref_q.QuantNLinear(m.in_features, m.out_features, m.bias is not None)
        else:
            continue
        q.weight = m.weight
        q.bias = m.bias
        swapped[id(m)] = q
        parent = model
        parts = name.split('.')
        for p in parts[:-1]:
            parent = getattr(parent, p)
        setattr(parent, parts[-1], q)
    for k in graph:
        if id(graph[k]) in swapped:
            graph[k] = swapped[id(graph[k])]
    return model


def _state(graph, targ):
    out = {}
    for i, k in enumerate(graph):
        m = graph[k]
        if type(m) in targ:
            out['L{}.w'.format(i)] = npy(m.weight)
            if m.bias is not None:
                out['L{}.b'.format(i)] = npy(m.bias)
        elif type(m) == nn.BatchNorm2d and hasattr(m, 'fake_weight'):
            out['L{}.fw'.format(i)] = npy(m.fake_weight)
            out['L{}.fb'.format(i)] = npy(m.fake_bias)
    return out


def _count_sweeps(ref_dfq, graph, rels, targ, **kw):
    """Run the reference's loop; count its per-sweep deepcopy of the graph (dfq.py:84) without touching its code."""
    n = {'n': 0}
    orig = copy.deepcopy

    def counting(x, *a, **k):
        if isinstance(x, dict) and 'Data' in x:
            n['n'] += 1
        return orig(x, *a, **k)
    copy.deepcopy = counting
    try:
        ref_dfq.cross_layer_equalization(graph, rels, targ, **kw)
    finally:
        copy.deepcopy = orig
    return n['n']


@pytest.mark.parametrize('name,seed,quant_layers,signed', [('tiny_mobile', 0, False, False), ('tiny_mobile', 1, True, False),
                                                           ('tiny_res', 0, True, False), ('tiny_cat', 3, False, True)])
def test_reference_prepared_graph_through_the_drop_in_entry_points(engine, ref, name, seed, quant_layers, signed, capsys):
    from dfq_amd import dfq, synthetic
    model, graph, bottoms = synthetic.build(name, seed=seed)
    if quant_layers:
        _swap_to_reference_layers(model, graph, ref.q)
        targ = [ref.q.QuantNConv2d, ref.q.QuantNLinear]
    else:
        targ = [nn.Conv2d, nn.Linear]
    # --- the reference prepares: main_cls.py:147-151
    model = ref.lt.merge_batchnorm(model, graph, bottoms, targ)
    rels = ref.rel.create_relation(graph, bottoms, targ, delete_single=False)
    assert rels and all(type(r) is ref.rel.Relation for r in rels)

    # one deep copy carries model, graph and relations together (shared module identities are kept by the memo)
    model_b, graph_b, rels_b = copy.deepcopy((model, graph, rels))
    assert all(type(r) is ref.rel.Relation for r in rels_b)
    keys = list(graph.keys())

    # --- the reference's own pass on its objects
    n_ref = _count_sweeps(ref.dfq, graph, rels, targ, converge_thres=2e-7, signed=signed)
    le_ref = _state(graph, targ)
    S_ref = [npy(r.get_scale_vec()) for r in rels]

    # --- the same objects through this package: same call, same keywords
    model_b.to(engine.device)
    dfq.cross_layer_equalization(graph_b, rels_b, targ, converge_thres=2e-7, signed=signed)
    assert dfq.last_equalization['sweeps'] == n_ref
    le_got = _state(graph_b, targ)
    assert set(le_got) == set(le_ref)
    for k in le_ref:
        assert_close(le_got[k], le_ref[k], '{} LE {}'.format(name, k))
    for i, (r, s) in enumerate(zip(rels_b, S_ref)):              # the REFERENCE's Relation objects carry the cumulative S
        assert_close(npy(r.get_scale_vec()), s, 'S of relation {}'.format(i))
        assert [keys.index(k) for k in r.get_idxs()] == [keys.index(k) for k in rels[i].get_idxs()]
    # the model sees the change: the graph's modules ARE the model's modules
    mods = {id(m) for m in model_b.modules()}
    assert all(id(graph_b[k]) in mods for k in graph_b if type(graph_b[k]) in targ)

    # --- bias correction, stage-wise: from the reference's post-LE state on both sides
    with torch.no_grad():
        for k in keys:
            a, b = graph[k], graph_b[k]
            if type(a) in targ:
                b.weight.copy_(a.weight)
                b.bias.copy_(a.bias)
            elif type(a) == nn.BatchNorm2d and hasattr(a, 'fake_weight'):
                b.fake_weight.copy_(a.fake_weight)
                b.fake_bias.copy_(a.fake_bias)
    ref.dfq.bias_correction(graph, bottoms, targ, signed=signed)
    dfq.bias_correction(graph_b, bottoms, targ, signed=signed)
    bc_ref, bc_got = _state(graph, targ), _state(graph_b, targ)
    for k in bc_ref:
        assert_close(bc_got[k], bc_ref[k], '{} BC {}'.format(name, k))
    capsys.readouterr()                                          # the reference prints progress lines
