"""torch.fx tracer that emits the reference's (graph, bottoms) dict pair.

The reference obtains these dicts from the third-party ``PyTransformer`` tracer
(main_cls.py:132-135; the submodule is not vendored, .gitmodules:1-3).  Only the *format* matters
to the calibration passes, and it is fixed by their call sites:

  * ``graph``   : OrderedDict key -> nn.Module | str, in forward order, first entry
                  ``graph['Data'] = 'Data'`` (relation.py:52, dfq.py:197);
  * ``bottoms`` : OrderedDict key -> list[key] | None (None only for 'Data');
  * tensor ops are string-valued entries whose key *contains* 'add', 'cat', 'F.pad',
    'torch.mean', 'F.interpolate' (layer_transform.py:316-326, relation.py:42-43).

Shape-only ops (view/size/flatten/getitem/contiguous/functional dropout in eval...) are transparent: their
consumers are wired to their producer, which is how the reference graphs look
(images/graph_cls.png ends ... ReLU -> torch.mean -> Linear).  Only a WHITELIST is transparent.
``nn.Dropout`` MODULES are nodes, as in the reference's graphs (images/graph_deeplab.png: Dropout_185,
Dropout_194, Dropout_198): relation.py:36-46 does not walk through them, so the layers either side of
a Dropout are never paired, and find_prev_bn's depth strings (layer_transform.py:337) count them.  Functional
activations the passes must see become module nodes (F.relu -> nn.ReLU(), F.relu6 -> nn.ReLU6(): relation.py:36-41
walks through ReLU but stops at ReLU6, dfq.py:209-211 and layer_transform.py:373-380 look for them behind a BN);
every other function or method (sigmoid, hardswish, mul, chunk, ...) is kept as an opaque string node, as the
reference's tracer does, so that `create_relation` / `find_prev_bn` stop there instead of pairing layers across
an operation that does not commute with a per-channel scale.
"""
from __future__ import annotations

import operator
from collections import OrderedDict

import torch
import torch.fx
import torch.nn as nn
import torch.nn.functional as F

_FUNC_NAMES = {
    operator.add: 'add', torch.add: 'add', operator.iadd: 'add',
    torch.cat: 'torch.cat', torch.mean: 'torch.mean',
    F.interpolate: 'F.interpolate', F.pad: 'F.pad', F.softmax: 'F.softmax',
}
_METHOD_NAMES = {'add': 'add', 'add_': 'add', 'mean': 'torch.mean'}
_TRANSPARENT_METHODS = {'view', 'reshape', 'flatten', 'contiguous', 'squeeze', 'unsqueeze',
                        'float', 'detach', 'clone'}
_TRANSPARENT_FUNCS = {torch.flatten, torch.reshape, torch.squeeze, torch.unsqueeze, operator.getitem,
                      F.dropout, F.dropout2d}          # models are traced in eval mode: dropout is the identity
# functional activations -> the module type the calibration passes test for
_FUNC_MODULES = {F.relu: nn.ReLU, torch.relu: nn.ReLU, F.relu6: nn.ReLU6}
_METHOD_MODULES = {'relu': nn.ReLU, 'relu_': nn.ReLU}
# results that are shapes / python numbers, not tensors: they must not become graph edges (e.g. the
# `size=(x.shape[2], x.shape[3])` argument of F.interpolate refers to x only for its shape)
_SHAPE_METHODS = {'size', 'dim', 'numel'}


class _LeafTracer(torch.fx.Tracer):
    """Every module that owns no sub-modules (or lives in torch.nn) is a leaf."""

    def is_leaf_module(self, m, qualname):
        if isinstance(m, (nn.Sequential, nn.ModuleList, nn.ModuleDict)):
            return False
        if m.__module__.startswith('torch.nn') or len(list(m.children())) == 0:
            return True
        return False


def _tensor_inputs(node):
    out = []

    def visit(a):
        if isinstance(a, torch.fx.Node):
            out.append(a)
        elif isinstance(a, (list, tuple)):
            for x in a:
                visit(x)
    for a in node.args:
        visit(a)
    for a in node.kwargs.values():
        visit(a)
    return out


def trace(model, key_style='name'):
    """Return (graph, bottoms) for ``model``.

    key_style 'name' keys module nodes as '<ClassName>_<index>' strings; 'module' keys them by
    the module object itself (any hashable works for the calibration passes).
    """
    graph, bottoms, _, _ = _trace(model, key_style)
    return graph, bottoms


def _trace(model, key_style='name'):
    tracer = _LeafTracer()
    fx_graph = tracer.trace(model)
    modules = dict(model.named_modules())

    graph = OrderedDict()
    bottoms = OrderedDict()
    graph['Data'] = 'Data'
    bottoms['Data'] = None
    alias = {}                    # fx node -> key (or the key of the producer it is transparent to)
    counter = 1

    def key_of(n):
        return alias[n]

    for n in fx_graph.nodes:
        if n.op == 'placeholder':
            alias[n] = 'Data'
            continue
        if n.op == 'output':
            continue
        ins = [i for i in _tensor_inputs(n) if i in alias]
        if n.op == 'call_module':
            m = modules[n.target]
            if isinstance(m, nn.Identity):
                alias[n] = key_of(ins[0])
                continue
            key = '{}_{}'.format(type(m).__name__, counter) if key_style == 'name' else m
            graph[key] = m
        elif n.op == 'call_function' and n.target is getattr:
            continue                                   # x.shape, x.dtype ...: not a tensor
        elif n.op == 'call_method' and n.target in _SHAPE_METHODS:
            continue
        elif n.op in ('call_function', 'call_method'):
            is_fn = n.op == 'call_function'
            names = _FUNC_NAMES if is_fn else _METHOD_NAMES
            mods = _FUNC_MODULES if is_fn else _METHOD_MODULES
            transparent = _TRANSPARENT_FUNCS if is_fn else _TRANSPARENT_METHODS
            if not ins:
                continue                                   # no tensor input: a constant, not a graph node
            if n.target in transparent:
                alias[n] = key_of(ins[0])
                continue
            if n.target in names:
                key = '{}_{}'.format(names[n.target], counter)
                graph[key] = key
            elif n.target in mods:
                m = mods[n.target]()
                key = '{}_{}'.format(type(m).__name__, counter) if key_style == 'name' else m
                graph[key] = m
            else:
                # unknown operation: an opaque node (a string, like the reference's tensor-op entries).  The passes
                # classify string nodes by SUBSTRINGS of the key ('add', 'cat', 'mean', 'pad', 'interpolate': relation.py:42-43,
                # layer_transform.py:316-326), so the key must not carry the operation's name (addmm, scatter, ...): the name
                # goes into the value
                name = getattr(n.target, '__name__', None) or str(n.target)
                key = 'opaque_{}'.format(counter)
                graph[key] = 'opaque:{}'.format(name if is_fn else 'Tensor.' + name)
        else:                      # get_attr etc.
            continue
        bots = []
        for i in ins:
            k = key_of(i)
            if k not in bots or n.op != 'call_module':
                bots.append(k)
        bottoms[key] = bots
        alias[n] = key
        counter += 1
    return graph, bottoms, fx_graph, alias


# the reference's list (utils/layer_transform.py:10-14): Tensor.__add__ / add / __iadd__, torch.cat, torch.mean,
# F.interpolate, F.softmax.  add / cat get one quantiser per input, the others one for their (first) input.
TENSOR_OPS = ('add', 'cat', 'mean', 'interpolate', 'softmax')


def quantize_tensor_ops(model, ops=TENSOR_OPS, num_bits=8, momentum=0.1, key_style='name'):
    """Activation quantisers on the inputs of tensor ops -- what the reference's ``switch_layers(quant_op=True)``
    + ``replace_op`` achieve by monkey-patching ``torch.Tensor.__add__`` / ``torch.cat`` / ``torch.mean`` /
    ``F.interpolate`` / ``F.softmax`` and routing their inputs through a ``CustomTensorOP`` container
    (layer_transform.py:16-228) -- done as a torch.fx rewrite: one ``QuantMeasure`` per tensor input of every add / cat
    node (one for the input of mean, interpolate, softmax) is inserted in front of the op.

    Returns ``(quantised GraphModule, graph, bottoms, tensor_op_quant)``.  ``graph`` / ``bottoms`` are those of
    ``trace(model)`` (the quantisers live on the edges, they are no graph nodes, as in the reference);
    ``tensor_op_quant`` maps the graph key of each rewritten op to its quantisers in input order and is what
    ``set_quant_minmax(..., tensor_op_quant=...)`` fills.  The GraphModule shares all layers with ``model``.
    """
    from .utils.quantize import QuantMeasure
    graph, bottoms, fx_graph, alias = _trace(model, key_style)
    holder = nn.ModuleList()
    tensor_op_quant = OrderedDict()
    root = model
    if hasattr(root, 'tensor_op_quant'):
        raise ValueError('model already has a tensor_op_quant container')
    root.add_module('tensor_op_quant', holder)
    for n in list(fx_graph.nodes):
        key = alias.get(n)
        if n.op not in ('call_function', 'call_method') or not isinstance(key, str) or graph.get(key) != key:
            continue
        if not any(tag in key for tag in ops):
            continue
        ins = [i for i in _tensor_inputs(n) if i in alias]
        if not ('add' in key or 'cat' in key):      # mean / interpolate / softmax: the data input only
            ins = ins[:1]
        quants = []
        replaced = {}
        for inp in ins:
            if inp in replaced:                 # x + x: one quantiser per distinct edge
                continue
            qm = QuantMeasure(num_bits=num_bits, momentum=momentum)
            holder.append(qm)
            with fx_graph.inserting_before(n):
                qn = fx_graph.call_module('tensor_op_quant.{}'.format(len(holder) - 1), (inp,))
            replaced[inp] = qn
            quants.append(qm)

        def swap(a):
            if isinstance(a, torch.fx.Node):
                return replaced.get(a, a)
            if isinstance(a, (list, tuple)):
                return type(a)(swap(x) for x in a)
            return a
        n.args = tuple(swap(a) for a in n.args)
        n.kwargs = {k: swap(v) for k, v in n.kwargs.items()}
        tensor_op_quant[key] = quants
    fx_graph.lint()
    gm = torch.fx.GraphModule(root, fx_graph)
    return gm, graph, bottoms, tensor_op_quant
