"""Graph-level passes around the equalisation/correction core, with the call surface of the
reference's ``utils/layer_transform.py``:

  merge_batchnorm      <- utils/layer_transform.py:231-276   (engine: dfq_fold_batchnorm)
  quantize_targ_layer  <- utils/layer_transform.py:279-296   (engine: dfq_quant_plan_*)
  find_prev_bn         <- utils/layer_transform.py:299-344   (host graph walk, O(#nodes))
  set_quant_minmax     <- utils/layer_transform.py:347-609   (engine: dfq_bn_ranges, dfq_relu_moments, ...)

The graph model is the reference's: ``graph`` maps key -> nn.Module | str (tensor ops are strings whose
key contains 'add' / 'cat' / ...), ``bottoms`` maps key -> list of input keys | None.
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn

from .. import _ffi
from .quantize import QConv2d, QuantConv2d, QuantNConv2d, QLinear, QuantLinear, QuantNLinear

_CONV_TYPES = (nn.Conv2d, QConv2d, QuantConv2d, QuantNConv2d)
_LINEAR_TYPES = (nn.Linear, QLinear, QuantLinear, QuantNLinear)


def _ensure_bias(layer):
    """The reference adds a zero bias Parameter when a layer has none (layer_transform.py:253-254)."""
    b = layer.__dict__['_parameters'].get('bias')        # (not layer.bias: Module.__getattr__ is the slow path, and table
    if b is not None:                                     # building asks thousands of layers)
        return b
    if layer.bias is None:
        layer.bias = nn.Parameter(torch.zeros(layer.weight.size(0), dtype=torch.float32,
                                              device=layer.weight.device), requires_grad=False)
    return layer.bias


def merge_batchnorm(model, graph, bottoms, targ_type=[QConv2d]):
    """Fold every BatchNorm2d that directly follows a targ layer into that layer.

    W <- W * gamma/sqrt(var+eps) per output channel, b <- b*gamma/sqrt(var+eps) + beta -
    gamma*mean/sqrt(var+eps); the BN keeps ``fake_weight = |gamma|`` and ``fake_bias = beta`` for the
    later passes and becomes an identity (gamma = var = 1, beta = mean = 0, eps below float32 resolution).
    """
    lib = _ffi.lib()
    with torch.no_grad():
        stage = _ffi.entry_stage()
        pairs = []
        for key in graph:
            bots = bottoms[key]
            if bots is None:
                continue
            bn = graph[key]
            if type(bn) != nn.BatchNorm2d:
                continue
            for bk in bots:
                layer = graph[bk]
                if type(layer) not in targ_type:
                    continue
                _ensure_bias(layer)
                pairs.append((layer, bn))
                break
        # a model that lives on the host crosses PCIe once each way: every tensor of every pair in one packed copy, the new
        # per-channel vectors of all BatchNorms in one flat buffer that comes back in one copy
        stage.prefetch([t for layer, bn in pairs for t in (layer.weight, layer.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)])
        fake = stage.new_flat(2 * sum(bn.weight.numel() for _, bn in pairs)) if pairs else None
        outs, at = [], 0
        for layer, bn in pairs:
            w = stage.bind(layer.weight)
            b = stage.bind(layer.bias)
            gamma, beta = stage.bind(bn.weight), stage.bind(bn.bias)
            mean, var = stage.bind(bn.running_mean), stage.bind(bn.running_var)
            n = gamma.numel()
            fw, fb = fake[at:at + n], fake[at + n:at + 2 * n]
            _ffi.check(lib.dfq_fold_batchnorm(_ffi.ptr(w), _ffi.ptr(b), w.shape[0], w[0].numel(), _ffi.ptr(gamma),
                                              _ffi.ptr(beta), _ffi.ptr(mean), _ffi.ptr(var),
                                              ctypes.c_float(bn.eps), _ffi.ptr(fw), _ffi.ptr(fb),
                                              _ffi.stream_arg()))
            outs.append((bn, at, n))
            at += 2 * n
            # The reference sets eps = 0 (layer_transform.py:272); current PyTorch rejects eps <= 0 in
            # F.batch_norm.  1e-12 is absorbed by float32 rounding (1 + 1e-12 == 1): the folded BN is still an
            # exact identity and the model still runs.
            bn.eps = 1e-12
        host = {}
        for bn, o, n in outs:
            dev = bn.weight.device
            if dev == fake.device:
                src = fake
            else:                                 # ONE copy of the flat buffer per foreign device (a CPU-resident model: one D2H)
                src = host.get(dev)
                if src is None:
                    src = host[dev] = _ffi._to_host(fake) if dev.type == 'cpu' else fake.to(dev)
            bn.register_buffer('fake_weight', src[o:o + n].clone())
            bn.register_buffer('fake_bias', src[o + n:o + 2 * n].clone())
        if fake is not None and any(bn.weight.device != fake.device for bn, _, _ in outs):
            # the proxies were computed on the device: a stage whose shadows outlive the call keeps `fake` as their device copy
            stage.adopt(fake, [(t, at, t.numel()) for bn, o, n in outs
                               for t, at in ((bn.fake_weight, o), (bn.fake_bias, o + n)) if t.device.type == 'cpu'])
        stage.writeback()
    return model


def quantize_targ_layer(graph, bit_weight=8, bits_bias=16, targ_type=None, return_codes=False):
    """Per-tensor asymmetric fake-quant of every targ layer's weight (and bias unless 32 bit).

    Two launches for the whole network: one multi-tensor min/max, one multi-tensor quantise.
    ``return_codes`` (extension) additionally returns {key: int32 code tensor of the weight}.
    """
    print("Quantizing Layer parameters")
    if bits_bias == 32:
        print("Skipping bias quantization (32 bits)")
    assert targ_type != None, "targ_type cannot be None!"
    lib = _ffi.lib()
    with torch.no_grad():
        stage = _ffi.entry_stage()
        segs, keep, codes = [], [], {}
        stage.prefetch([t for layer in graph.values() if type(layer) in targ_type for t in (layer.weight, layer.bias)])
        for key in graph:
            layer = graph[key]
            if type(layer) not in targ_type:
                continue
            w = stage.bind(layer.weight)
            c = None
            if return_codes:
                c = stage.new(w.shape, dtype=torch.int32)
                codes[key] = c
            segs.append(_ffi.DfqSegment(w.data_ptr(), w.numel(), int(bit_weight), 0, c.data_ptr() if c is not None else None))
            keep.append(w)
            if layer.bias is not None and bits_bias < 32:
                b = stage.bind(layer.bias)
                segs.append(_ffi.DfqSegment(b.data_ptr(), b.numel(), int(bits_bias), 0, None))
                keep.append(b)
        if segs:
            arr = (_ffi.DfqSegment * len(segs))(*segs)
            plan = ctypes.c_void_p()
            _ffi.check(lib.dfq_quant_plan_create(arr, len(segs), ctypes.byref(plan)))
            try:
                _ffi.check(lib.dfq_quant_plan_run(plan, _ffi.stream_arg()))
                _ffi.synchronize()
            finally:
                lib.dfq_quant_plan_destroy(plan)
        stage.writeback()
    if return_codes:
        return graph, codes
    return graph


def find_prev_bn(bn_module, relu_attached, graph, bottoms, bot):
    """Breadth-first walk upwards from ``bot`` to the nearest BatchNorm on every path.

    Returns (bn_list, relu_attach_list, connect_type_list, targ_without_bn) like
    layer_transform.py:299-344: ``bn_list`` holds (bn module, branch id) where the branch id is a
    string of length = depth of the hit; the connect type ('one' / 'add' / 'add_<flag>' / 'cat')
    is inherited from the last add/cat node passed on the way up.
    """
    frontier = [(b, str(i)) for i, b in enumerate(bot)]
    ctype = {str(i): 'one' for i in range(len(bot))}
    bn_list, relu_list, connect_list = [], [], []
    no_bn_targets = {}
    merged = False
    while frontier:
        key, bid = frontier.pop(0)
        node = graph[key]
        if isinstance(node, str):
            if 'add' in key:
                ctype[bid] = 'add_{}'.format(relu_attached[key]) if key in relu_attached else 'add'
                merged = True
            elif 'cat' in key:
                ctype[bid] = 'cat'
                merged = True
        elif not merged and type(node) in _CONV_TYPES + _LINEAR_TYPES:
            print("Warning: {} layer before first batch norm layer detected. The calculated value range might be off.".format(type(node)))
            assert bid[0] not in no_bn_targets, "Multiple conv/linear layer without batch_norm is not supported."
            no_bn_targets[bid[0]] = ("conv" if type(node) in _CONV_TYPES else "linear", node)
        if key in bn_module:
            bn_list.append((bn_module[key], bid))
            relu_list.append(relu_attached[key])
            connect_list.append(ctype[bid])
        else:
            deeper = bid + bid[0]
            frontier.extend((up, deeper) for up in bottoms[key])
            ctype[deeper] = ctype[bid]
    return bn_list, relu_list, connect_list, no_bn_targets


# ------------------------------------------------------------------------------------------------
# set_quant_minmax (layer_transform.py:347-609)
# ------------------------------------------------------------------------------------------------
_RELU_MODE = {'none': 0, 'relu': 1, 'relu6': 2}


class _Moments:
    """(mean, var) channel vectors of one branch, living on the device (layer_transform.py:494-540)."""

    def __init__(self, stage, n):
        self.stage = stage
        self.n = n
        self.mean = stage.new((n,))
        self.var = stage.new((n,))

    def add_source(self, bn, relu, accumulate):
        lib = _ffi.lib()
        w, b = self.stage.bind(bn.fake_weight), self.stage.bind(bn.fake_bias)
        _ffi.check(lib.dfq_relu_moments(_ffi.ptr(w), _ffi.ptr(b), self.n, _RELU_MODE[relu], _ffi.ptr(self.mean),
                                        _ffi.ptr(self.var), int(accumulate), _ffi.stream_arg()))

    def relu_after_add(self, mode, eps):
        _ffi.check(_ffi.lib().dfq_moments_after_add(_ffi.ptr(self.mean), _ffi.ptr(self.var), self.n, mode, eps,
                                                    _ffi.stream_arg()))

    def value_range(self, eps, n_sigma):
        out = self.stage.new((2,))
        _ffi.check(_ffi.lib().dfq_moment_range(_ffi.ptr(self.mean), _ffi.ptr(self.var), self.n, eps, float(n_sigma),
                                               _ffi.ptr(out), _ffi.stream_arg()))
        lo, hi = out.tolist()
        return lo, hi


def _bn_ranges(stage, reqs, n_sigma):
    """[(fake_weight, fake_bias, relu)] -> [(min, max)] with the ReLU clamps: ONE launch, one read-back."""
    if not reqs:
        return []
    lib = _ffi.lib()
    arr = (_ffi.DfqBnRangeReq * len(reqs))()
    keep = []
    for i, (fw, fb, relu) in enumerate(reqs):
        w, b = stage.bind(fw).reshape(-1), stage.bind(fb).reshape(-1)
        keep.append((w, b))
        arr[i] = _ffi.DfqBnRangeReq(w.data_ptr(), b.data_ptr(), w.numel(), _RELU_MODE[relu])
    out = stage.new((len(reqs), 2))
    scratch = stage.new((int(lib.dfq_bn_ranges_scratch_bytes(len(reqs))) // 4 + 1,), dtype=torch.int32)
    _ffi.check(lib.dfq_bn_ranges(arr, len(reqs), float(n_sigma), _ffi.ptr(out), _ffi.ptr(scratch), _ffi.stream_arg()))
    return [tuple(r) for r in out.tolist()]


def _through_layer(stage, layer, kind, vec):
    """A BN proxy vector through a conv / linear layer that has no BN of its own (case d, :455-463)."""
    lib = _ffi.lib()
    w = stage.bind(layer.weight)
    khkw = w[0, 0].numel() if w.dim() == 4 else 1
    out = stage.new((w.shape[0],))
    _ffi.check(lib.dfq_bn_through_layer(_ffi.ptr(w), w.shape[0], w.shape[1], khkw, getattr(layer, 'groups', 1) if kind == 'conv' else 1,
                                        _ffi.ptr(stage.bind(layer.bias)), _ffi.ptr(stage.bind(vec).reshape(-1)), _ffi.ptr(out),
                                        _ffi.stream_arg()))
    return out


def set_quant_minmax(graph, bottoms, is_detection=False, bn_type=torch.nn.BatchNorm2d, N=6, verbose=True,
                     tensor_op_quant=None):
    """Set ``running_min`` / ``running_max`` of every activation quantiser from the statistics of the
    BatchNorm layers in front of it (layer_transform.py:347-609); no data involved.

    Quantisers are the ``.quant`` modules of the Q*Conv2d / Q*Linear layers.  The reference additionally
    serves quantisers of tensor ops (QuantAdd, ...) that its ``replace_op`` registers in a module global
    of the PyTransformer machinery; a caller that has such modules passes them as
    ``tensor_op_quant = {graph key of the op: [QuantMeasure, ...]}``.  Same cases as the reference:
    1 BN -> 1 quantiser (:444-474), 1 quantiser fed by several BNs through add / cat (:476-580), several
    quantisers of one tensor op (:582-601), and a conv / linear without BN in between (case d).
    """
    if verbose:
        print("SET QUANT MIN MAX")
    eps = 1e-6
    bn_module, relu_attached = {}, {}
    stage = _ffi.entry_stage()
    one_to_one = []          # (quantiser, fake_weight, fake_bias, relu) resolved with one launch at the end
    with torch.no_grad():
        for key in graph:
            bot = bottoms[key]
            if bot is None:
                continue
            layer = graph[key]
            if type(layer) == bn_type:
                bn_module[key] = layer
                relu_attached[key] = 'none'
                continue
            if type(layer) == torch.nn.ReLU:
                relu_attached[bot[0]] = 'relu'
            elif type(layer) == torch.nn.ReLU6:
                relu_attached[bot[0]] = 'relu6'
            if isinstance(layer, str):
                quant_module = (tensor_op_quant or {}).get(key)
            elif hasattr(layer, 'quant'):
                quant_module = [layer.quant]
            else:
                quant_module = None
            if len(bot) == 1 and bot[0] == 'Data':
                if quant_module is None:
                    continue
                if is_detection:
                    quant_module[0].running_max.fill_(1)
                    quant_module[0].running_min.fill_(-1)
                else:                                           # (x - mean) / std of the data pre-processing
                    quant_module[0].running_max.fill_(2.64)
                    quant_module[0].running_min.fill_(-2.11790393)
                continue
            if quant_module is None:
                continue
            bn_list, relu_list, connect_list, no_bn = find_prev_bn(bn_module, relu_attached, graph, bottoms, bot[:])
            if len(quant_module) == len(bn_list):               # 1 to 1 mapping
                for q, (bn, bid), relu in zip(quant_module, bn_list, relu_list):
                    if bid[0] in no_bn:                         # case (d): no ReLU clamp in the reference
                        kind, obj = no_bn[bid[0]]
                        fb = _through_layer(stage, obj, kind, bn.fake_bias)
                        fw = _through_layer(stage, obj, kind, bn.fake_weight)
                        one_to_one.append((q, fw, fb, 'none'))
                    else:
                        one_to_one.append((q, bn.fake_weight, bn.fake_bias, relu))
                continue
            # ---- 1 to many / many to many ----
            branches = {}
            for ent, relu, ctype in zip(bn_list, relu_list, connect_list):
                branches.setdefault(ent[1][0], []).append((ent, relu, ctype))
            results = {}
            for bkey, items in branches.items():
                items = sorted(items, key=lambda x: len(x[0][1]), reverse=True)
                (bn, bid), use_relu, connect_type = items.pop(0)
                depth = len(bid)
                mom = None
                value_min = value_max = None
                if 'add' in connect_type:
                    mom = _Moments(stage, bn.fake_bias.numel())
                    mom.add_source(bn, use_relu, accumulate=False)
                else:
                    (value_min, value_max), = _bn_ranges(stage, [(bn.fake_weight, bn.fake_bias, use_relu)], N)
                while items:
                    bound = 0
                    while bound < len(items) and len(items[bound][0][1]) == depth:
                        bound += 1
                    if bound == 0:
                        depth = len(items[0][0][1])             # cut depth
                        continue
                    for (bn, bid), relu_t, connect_type in items[:bound]:
                        if 'add' in connect_type:
                            mom.add_source(bn, relu_t, accumulate=True)
                            if 'relu6' in connect_type:
                                mom.relu_after_add(2, eps)
                            elif 'relu' in connect_type:
                                mom.relu_after_add(1, eps)
                        elif connect_type == 'cat':
                            (lo, hi), = _bn_ranges(stage, [(bn.fake_weight, bn.fake_bias, relu_t)], N)
                            value_min = min(value_min, lo)
                            value_max = max(value_max, hi)
                        else:
                            # `if use_relu_tmp` of the reference is always true (a non-empty string): clamp at 0
                            (lo, hi), = _bn_ranges(stage, [(bn.fake_weight, bn.fake_bias, 'none')], N)
                            value_min += max(0., lo)
                            value_max += hi
                    items = items[bound:]
                    if connect_type == 'one':
                        value_min /= (bound + 1)
                        value_max /= (bound + 1)
                if 'add' in connect_type:
                    results[bkey] = mom.value_range(eps, N)
                else:
                    results[bkey] = (value_min, value_max)
            if len(quant_module) == 1 and len(quant_module) < len(bn_list):     # 1 to many
                assert len(results) == 1, "Error occurs when setting min/max, should be 1 to many"
                value_min, value_max = list(results.values())[0]
                quant_module[0].running_max.fill_(value_max)
                quant_module[0].running_min.fill_(value_min)
            elif len(quant_module) < len(bn_list):                              # many to many
                assert len(results) == len(quant_module), 'LENGTH NOT EQUAL {} vs {}'.format(len(results), len(quant_module))
                for idx, q in enumerate(quant_module):
                    value_min, value_max = results[str(idx)]
                    q.running_max.fill_(value_max)
                    q.running_min.fill_(value_min)
            else:
                assert False, "Unknown error occured while setting min/max"
        ranges = _bn_ranges(stage, [(fw, fb, relu) for (_, fw, fb, relu) in one_to_one], N)
        for (q, _, _, _), (lo, hi) in zip(one_to_one, ranges):
            q.running_max.fill_(hi)
            q.running_min.fill_(lo)
