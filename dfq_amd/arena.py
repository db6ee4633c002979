"""A batch of networks of ONE architecture laid out at a fixed stride in one device allocation.

The calibration of a batch binds some 220 tensors per network (weights, biases, the BatchNorm proxies fake_weight /
fake_bias, the cumulative scale vectors): 7 000 addresses for the benchmark's batch of 32, and gathering them -- one
``data_ptr()`` / contiguity / device check per tensor, in Python -- was 20 of the 23 ms that building the two plans of a
batch cost, against 7.6 ms of GPU work for the whole calibration.  ``NetworkBatch`` moves the tensors ONCE, when the batch
is put together (model loading, not calibration), into one allocation in which network n's tensors sit at the same offsets
from ``base + n * stride``; the modules' parameters and buffers are re-pointed at those slots (views, so the models keep
working as before; ``release()`` gives them storages of their own again -- do that before saving or deep-copying a packed
model, because torch pickles and copies a view together with its whole storage).  A plan over the batch is then the tables of the FIRST network plus one base address per network
(``dfq_le_plan_create_replicated`` / ``dfq_bc_plan_create_replicated``, include/dfq_hip.h): no per-tensor host work at all.

This is the host side of the reference's per-network graph walks (dfq.py:78-82, :194-270) for a batch; the arithmetic is the
engine's, unchanged -- the plans a NetworkBatch creates are the plans ``build_le_plan_batch`` / ``build_bc_plan_batch``
would create over the same tensors (tests/test_arena.py).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _ffi
from . import dfq as _dfq
from .utils.layer_transform import _ensure_bias

_ALIGN = 64            # floats: every tensor starts on a 256-byte boundary (vector loads, the alignment hipMalloc gives)


def _rebind(mod, name, view):
    """point a module's parameter / buffer / plain tensor attribute at `view` (same values, new storage)"""
    prm = mod.__dict__['_parameters']
    if prm.get(name) is not None:
        prm[name].data = view
    elif name in mod.__dict__['_buffers']:
        mod.__dict__['_buffers'][name] = view
    else:
        setattr(mod, name, view)


class NetworkBatch:
    """``nets``: list of (graph, bottoms, relations) of networks with the same graph (keys, node types, tensor shapes,
    relation triples), BatchNorm already folded (``merge_batchnorm``) and relations created, tensors float32 on the target
    device.  After construction every tensor the equalisation and the bias correction touch lives in ``self.storage``."""

    def __init__(self, nets, targ_type, bn_type=torch.nn.BatchNorm2d, stage=None):
        if not nets:
            raise ValueError('NetworkBatch: no networks')
        self.stage = stage or _ffi.Stage()
        dev = self.stage.device
        self.nets = list(nets)
        self.targ_type, self.bn_type = targ_type, bn_type
        g0, b0, r0 = self.nets[0]
        le_t = _dfq._le_template(g0, r0, targ_type)
        bc_t = _dfq._bc_template(g0, b0, targ_type, bn_type)
        need_bias = sorted(set(le_t['firsts']) | set(bc_t['bias_layers']))
        # (graph key, attribute) of every BatchNorm proxy a table refers to, in first-use order
        bn_refs, seen = [], set()

        def add(ref):
            if ref is not None and ref not in seen:
                seen.add(ref)
                bn_refs.append(ref)
        for (_, _, kb) in le_t['rel']:
            if kb is not None:
                add((kb, 'fake_weight'))
                add((kb, 'fake_bias'))
        for _, ref in bc_t['step_next']:
            add(ref)
        for _, ref in bc_t['src_fw']:
            add(ref)
        for _, ref in bc_t['src_fb']:
            add(ref)

        def slots_of(graph, bottoms, relations):
            """[(setter, tensor or None, numel)] of one network in slot order; None = a scale vector still to be created"""
            if _dfq._le_template(graph, relations, targ_type) is not le_t or _dfq._bc_template(graph, bottoms, targ_type, bn_type) is not bc_t:
                raise ValueError('NetworkBatch: the networks of a batch must share one architecture (graph keys, node types, '
                                 'tensor shapes, relations)')
            mods = [graph[k] for k in le_t['keys']]
            for i in need_bias:
                _ensure_bias(mods[i])                                   # dfq.py:91-92, layer_transform.py:253-254
            out = []
            for m in mods:
                prm = m.__dict__['_parameters']
                out.append((m, 'weight', prm['weight']))
                if prm.get('bias') is not None:
                    out.append((m, 'bias', prm['bias']))
            for (key, name) in bn_refs:
                t = _dfq._attr(graph[key], name)
                if t is not None:
                    out.append((graph[key], name, t))
            for rr, o1 in zip(relations, le_t['o1']):
                out.append((rr, 'S', rr.S if rr.S is not None else o1))
            return out

        per_net = [slots_of(*net) for net in self.nets]
        numel = [t.numel() if torch.is_tensor(t) else int(t) for (_, _, t) in per_net[0]]
        for n, slots in enumerate(per_net):
            if [t.numel() if torch.is_tensor(t) else int(t) for (_, _, t) in slots] != numel:
                raise ValueError('NetworkBatch: network {} does not have the tensors of network 0'.format(n))
            for (_, name, t) in slots:
                if torch.is_tensor(t) and (t.dtype is not torch.float32 or t.device != dev):
                    raise ValueError('NetworkBatch: {} of network {} is {} on {}; the batch wants float32 on {}'.format(
                        name, n, t.dtype, t.device, dev))
        offs, total = [], 0
        for c in numel:
            offs.append(total)
            total += -(-c // _ALIGN) * _ALIGN
        self.stride = total                                            # floats per network
        self.storage = torch.empty(len(self.nets) * total, dtype=torch.float32, device=dev)
        rows = self.storage.view(len(self.nets), total)
        srcs, dsts = [], []
        with torch.no_grad():
            for n, slots in enumerate(per_net):
                row = rows[n]
                for (owner, name, t), off, c in zip(slots, offs, numel):
                    if torch.is_tensor(t):
                        view = row[off:off + c].view(t.shape)
                        srcs.append(t.detach())
                        dsts.append(view)
                    else:
                        view = row[off:off + c]
                        view.fill_(1.0)                                # Relation.S starts at 1 (relation.py:11)
                    if name == 'S':
                        owner.S = view
                    else:
                        _rebind(owner, name, view)
            if srcs:
                torch._foreach_copy_(dsts, srcs)
        self.slots_per_network = len(numel)
        self.bases = (np.uint64(self.storage.data_ptr()) + np.arange(len(self.nets), dtype=np.uint64) * np.uint64(4 * total))
        # the first network's tables (absolute addresses inside its slot); every plan of this batch starts from them
        self._le = _dfq._fast_le_tables([(g0, r0)], targ_type, dev)
        self._bc = _dfq._fast_bc_tables([(g0, b0)], targ_type, bn_type, dev)
        lo, hi = int(self.bases[0]), int(self.bases[0]) + 4 * total
        for T in (self._le, self._bc):
            if T is None:
                raise RuntimeError('NetworkBatch: the tables of the first network could not be built')
            for a in T.arrays.values():
                for field in (a.dtype.names or ()):
                    if a.dtype[field] == np.uint64:
                        p = a[field][a[field] != 0]
                        if len(p) and (p.min() < lo or p.max() >= hi):
                            raise RuntimeError('NetworkBatch: a table of the first network points outside its slot ({})'.format(field))
        self._base_ints = [int(v) for v in self.bases]
        self._scale_cum = [rr.S for (_, _, rels) in self.nets for rr in rels]
        self._probe = [(slots[0][2], slots[-1][0]) for slots in per_net]       # first weight, last relation of every network

    def release(self):
        """Give every tensor a storage of its own again (a copy of its slot) and drop the batch allocation.  The models' tensors
        are VIEWS of ``self.storage`` while the batch exists, and torch treats a view as its whole storage when it pickles or
        deep-copies one: ``torch.save(model.state_dict())`` / ``copy.deepcopy(model)`` of a packed model would carry all the
        batch's networks.  Call this when the calibration is done and the models go their own ways; plans created from the batch
        must not be run afterwards."""
        home = self.storage.untyped_storage().data_ptr()

        def mine(t):
            return torch.is_tensor(t) and t.untyped_storage().data_ptr() == home
        with torch.no_grad():
            for (graph, bottoms, relations) in self.nets:
                for m in graph.values():
                    if not isinstance(m, torch.nn.Module):
                        continue                                    # functional nodes of the graph are recorded by name
                    for name, t in m.__dict__['_parameters'].items():
                        if mine(t):
                            t.data = t.data.clone()
                    bufs = m.__dict__['_buffers']
                    for name in list(bufs):
                        if mine(bufs[name]):
                            bufs[name] = bufs[name].clone()
                    for name, t in list(m.__dict__.items()):
                        if mine(t):
                            m.__dict__[name] = t.clone()
                for rr in relations:
                    if mine(rr.S):
                        rr.S = rr.S.clone()
        self._probe, self._scale_cum = [], []
        self.storage = None

    # -- plans ---------------------------------------------------------------------------------------------------------
    def _tables(self, T):
        out = _dfq._Tables()
        out.arrays = T.arrays
        out.n_layers = T.n_layers
        out.keep = [self]
        out.bases = self.bases
        out.mutable = [self.storage]          # one allocation holds every tensor the plans rewrite: a snapshot is ONE copy
        return out

    def check(self, thorough=False):
        """Raise if a tensor has left its slot (someone assigned a new tensor to ``weight.data`` or ``Relation.S`` after the
        batch was put together).  The quick form looks at the first and the last slot of every network."""
        if self.storage is None:
            raise RuntimeError('NetworkBatch: the batch has been released')
        span = 4 * self.stride
        for n, ((w, rr), base) in enumerate(zip(self._probe, self._base_ints)):
            if w.data_ptr() != base or rr.S is None or not (base <= rr.S.data_ptr() < base + span):
                raise RuntimeError('NetworkBatch: a tensor of network {} no longer lives in the batch allocation'.format(n))
        if thorough:
            for n, (g, b, r) in enumerate(self.nets):
                le = _dfq._fast_le_tables([(g, r)], self.targ_type, self.stage.device)
                bc = _dfq._fast_bc_tables([(g, b)], self.targ_type, self.bn_type, self.stage.device)
                for T, T0 in ((le, self._le), (bc, self._bc)):
                    for key, a in T.arrays.items():
                        for field in (a.dtype.names or ()):
                            if a.dtype[field] == np.uint64:
                                want = T0.arrays[key][field].copy()
                                want[want != 0] += self.bases[n] - self.bases[0]
                                if not np.array_equal(a[field], want):
                                    raise RuntimeError('NetworkBatch: {} of network {} is not where the batch put it'.format(field, n))

    def le_plan(self):
        """One equalisation plan over the whole batch (LEPlan): the first network's tables + a base address per network."""
        self.check()
        t = self._tables(self._le)
        t.n_relations, t.scale_cum = self._le.n_relations, self._scale_cum
        return _dfq.LEPlan(t, None, stage=self.stage)

    def bc_plan(self):
        """One bias-correction plan over the whole batch (BCPlan)."""
        self.check()
        t = self._tables(self._bc)
        t.n_steps, t.n_sources = self._bc.n_steps, self._bc.n_sources
        t.step_out_ch, t.step_in = self._bc.step_out_ch, self._bc.step_in
        return _dfq.BCPlan(t, None, stage=self.stage)
