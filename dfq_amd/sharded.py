"""Cross-layer equalisation of ONE network sharded over several GPUs.

Channels of a relation are independent, relations that share a layer are order-dependent, so the
shard unit is a connected component of the relation graph (MobileNetV2: 16 components).  Every rank
holds the whole network.  A rank runs the sequential sweeps of dfq.py:85-101 on *scratch copies* of
the components it owns -- all it keeps of that is the cumulative scale vector of every owned relation
(utils/relation.py:20-24), bit-identical to the single-GPU loop's -- then ONE ``all_gather`` (RCCL
over xGMI when the process group is 'nccl') exchanges those vectors: 4 bytes per paired channel,
64 KB for MobileNetV2.  After it EVERY rank, the owner of a component included, rebuilds every paired
tensor from its pristine value with one batched engine launch,

    W = diag(S_out) . W0 . diag(1 / S_in),    b = b0 . S_out,    BN proxies likewise

(``dfq_rebuild_plan_run``: per element fl(fl(w0 * s_out) / s_in)).  All ranks execute the same launch
on the same inputs, so they end with **bit-identical networks** -- and the same network for every
world size, N = 1 included -- which is what lets the bias-correction chain and the int8 quantisation
that follow be plain replicas.  The result is within 1e-5 (relative) of the sequentially rescaled
tensors of the reference loop, the contract of BASELINE.json; the scale vectors themselves are
bit-identical to it.

This is a latency-bound exchange on a millisecond-scale job: it exists for configuration 4 of
BASELINE.json (DeepLab sharded over 8 GPUs) and for networks too large for one pass to be cheap, not
for the headline MobileNetV2 number (bench.py shards whole networks over ranks instead).

The data-dependent convergence test of dfq.py:83-115 needs the sum of all layers' mean |dW|: with
``max_sweeps=None`` the ranks run CHUNKS of sweeps, all-reduce a chunk's per-sweep sums in one collective and
draw the verdicts on the device (``ShardedEqualizer._run_data_dependent``): one host read per chunk.
"""
from __future__ import annotations

import copy
import ctypes
from collections import OrderedDict

import torch
import torch.distributed as dist

from . import _ffi
from . import dfq as _dfq
from .utils.layer_transform import _ensure_bias


def relation_components(relations):
    """Connected components of the relation list (relations sharing a layer are connected).
    Returns a list of lists of relation indices, each in list (= Gauss-Seidel) order."""
    parent = {}

    def find(x):
        while parent.setdefault(x, x) != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for rr in relations:
        a, b, _ = rr.get_idxs()
        parent[find(('L', a))] = find(('L', b))
    comps = {}
    for i, rr in enumerate(relations):
        comps.setdefault(find(('L', rr.get_idxs()[0])), []).append(i)
    return list(comps.values())


def assign_components(graph, relations, world_size):
    """Greedy bin packing of components by paired elements.  Deterministic -> identical on every rank."""
    comps = relation_components(relations)

    def cost(c):
        return sum(graph[relations[i].get_idxs()[0]].weight.numel() + graph[relations[i].get_idxs()[1]].weight.numel()
                   for i in c)
    order = sorted(range(len(comps)), key=lambda k: (-cost(comps[k]), comps[k][0]))
    load = [0] * world_size
    owner = [0] * len(relations)
    for k in order:
        r = min(range(world_size), key=lambda q: (load[q], q))
        load[r] += cost(comps[k])
        for i in comps[k]:
            owner[i] = r
    return owner


def _khkw(w):
    n = 1
    for d in w.shape[2:]:
        n *= int(d)
    return n


class _RebuildPlan:
    """ONE launch over a table of (src, dst, s_out, s_in) items -- ``dfq_rebuild_plan_*`` of include/dfq_hip.h."""

    def __init__(self, items):
        """items: list of (src, dst, s_out|None, s_in|None, groups) device tensors; vectors have groups == 1."""
        self._keep = items
        arr = (_ffi.DfqRebuildItem * len(items))()
        for a, (src, dst, so, si, groups) in zip(arr, items):
            a.src, a.dst = src.data_ptr(), dst.data_ptr()
            a.s_out = so.data_ptr() if so is not None else None
            a.s_in = si.data_ptr() if si is not None else None
            a.rows = int(src.shape[0])
            a.cols = int(src.shape[1]) if src.dim() > 1 else 1
            a.khkw = _khkw(src)
            a.groups = int(groups)
            a.in_reciprocal = 0
        self._plan = ctypes.c_void_p()
        _ffi.check(_ffi.lib().dfq_rebuild_plan_create(arr, len(items), ctypes.byref(self._plan)))

    @property
    def elements(self):
        return int(_ffi.lib().dfq_rebuild_plan_elements(self._plan))

    def run(self):
        _ffi.check(_ffi.lib().dfq_rebuild_plan_run(self._plan, _ffi.stream_arg()))

    def close(self):
        if self._plan:
            _ffi.lib().dfq_rebuild_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _torch_rebuild(src, dst, s_out, s_in, groups):
    """The same two roundings with torch ops (stand-in for the engine launch in the CPU-only logic tests)."""
    t = src
    if s_out is not None:
        t = t * s_out.view((-1,) + (1,) * (t.dim() - 1))
    if s_in is not None:
        per_row = s_in.view(groups, -1).repeat_interleave(src.shape[0] // groups, dim=0)
        t = t / per_row.view((src.shape[0], -1) + (1,) * (src.dim() - 2))
    dst.copy_(t)


class _EngineSession:
    """The owned relations of this rank as ONE engine plan over scratch copies of the owned tensors.  The plan's
    cumulative-scale buffers ARE the rank's slice of the exchange buffer.  The device-side loop state is configured
    never to stop by itself (threshold -1, no sweep cap): the stopping rule of dfq.py:83-115 needs the sum over ALL
    ranks and is evaluated by the caller."""

    def __init__(self, layers, rels, s_range, signed, eps):
        self.cfg = dict(s_range=tuple(s_range), signed=signed, eps=eps, converge_thres=-1.0, converge_count=10 ** 9,
                        max_sweeps=None)
        self.plan = _dfq.LEPlan(layers, rels)

    def start(self):
        self.plan.enqueue(0, restart=True, **self.cfg)       # statistics of the (just snapshotted) scratch tensors

    def sweeps(self, n):
        self.plan.enqueue(int(n), restart=False, **self.cfg)

    def last_diff(self):
        """sum over the owned layers of mean|W - W_prev| of the latest sweep (one small device-to-host copy)."""
        return self.plan.query()['last_diff_tmp']

    def set_log(self, log):
        """From now on diff_tmp of sweep j (since the last restart) is also left in log[j] (dfq_le_set_diff_log)."""
        _ffi.check(_ffi.lib().dfq_le_set_diff_log(self.plan._plan, log.data_ptr() if log is not None else None,
                                                  int(log.numel()) if log is not None else 0))

    def finish(self):
        self.plan.query()                                    # synchronises and surfaces a failed in-launch wait

    def close(self):
        self.plan.close()


class _RunnerSession:
    """Adapter for an injected ``le_runner`` (tests): every call equalises the scratch graph for n more sweeps; the
    cumulative scales it leaves in ``Relation.S`` are copied into the exchange buffer."""

    def __init__(self, runner, graph, relations, targ_type, s_range, signed, eps, scale_slices):
        self.relations, self.scale_slices = relations, scale_slices
        self.saved = [rr.S for rr in relations]
        for rr in relations:
            rr.S = None
        self.run = lambda n: runner(graph, relations, targ_type, max_sweeps=n, converge_thres=-1.0,
                                    converge_count=10 ** 9, s_range=list(s_range), signed=signed, eps=eps)
        self.res = None

    def start(self):
        pass

    def sweeps(self, n):
        self.res = self.run(int(n))

    def last_diff(self):
        return self.res['last_diff_tmp']

    def finish(self):
        for rr, dst, old in zip(self.relations, self.scale_slices, self.saved):
            if rr.S is not None:
                dst.copy_(rr.S)
            rr.S = old

    def close(self):
        pass


class ShardedEqualizer:
    """One network, one rank's share of it.  Construction (host side, untimed in bench.py) partitions the relation
    graph and builds the tables: the engine plan of the owned components over scratch tensors, the flat exchange
    buffer the plan accumulates its scales into, the snapshot and rebuild launches.  ``run`` is the data path:
    snapshot of the owned tensors (1 launch) -> local sweeps -> ONE all_gather -> rebuild of every paired tensor on
    every rank (1 launch).

    SINGLE USE: ``run`` may be called once.  The rebuild is in place (the pristine tensors are its source AND its
    destination), the exchange buffer starts at 1 and the sweeps accumulate into it, and ``check`` closes the sweep plan --
    a second ``run`` would apply the first run's scales again.  It raises instead; build a new equalizer for another pass
    (bench.py does, untimed)."""

    def __init__(self, graph, relations, targ_type, group=None, s_range=(1e-8, 1e8), signed=False, eps=0,
                 le_runner=None, use_torch_rebuild=False):
        self.graph, self.relations, self.targ_type, self.group = graph, relations, targ_type, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.owner = assign_components(graph, relations, self.world)
        self.stage = _ffi.Stage()
        dev = self.dev = self.stage.device
        self.comm_dev = dev if dist.get_backend(group) == 'nccl' else torch.device('cpu')
        self._torch_rebuild = use_torch_rebuild
        self._ran = False
        first_of, second_of = {}, {}
        with torch.no_grad():
            for i, rr in enumerate(relations):
                a, b, _ = rr.get_idxs()
                assert a not in first_of and b not in second_of, 'a layer is the first / second layer of one relation at most'
                first_of[a], second_of[b] = i, i
                _ensure_bias(graph[a])                                   # dfq.py:91-92, on every rank

            # ---- exchange layout: the cumulative S of every relation in list order (4 B per paired channel) ----
            self.lens = [graph[rr.get_idxs()[0]].weight.size(0) for rr in relations]
            self.offsets = [0]
            for n in self.lens:
                self.offsets.append(self.offsets[-1] + n)
            total = self.total = self.offsets[-1]
            self.exchange_bytes = 4 * total
            self.flat = torch.ones(max(1, total), dtype=torch.float32, device=dev)
            self.gathered = torch.ones(self.world * max(1, total), dtype=torch.float32, device=dev)
            gat = self.gathered.view(self.world, -1)
            s_of = [gat[o, self.offsets[i]:self.offsets[i + 1]] for i, o in enumerate(self.owner)]   # fixed addresses
            sel = torch.cat([torch.arange(self.offsets[i], self.offsets[i + 1]) + o * max(1, total)
                             for i, o in enumerate(self.owner)]) if relations else torch.zeros(0, dtype=torch.int64)
            self._sel = sel.to(dev)

            # ---- the tensors the rebuild rewrites, bound once (CPU-resident models get device shadows) ----
            bind = self.stage.bind
            paired = [k for k in graph if k in first_of or k in second_of]
            items = []
            for k in paired:
                layer = graph[k]
                w = bind(layer.weight)
                so = s_of[first_of[k]] if k in first_of else None
                si = s_of[second_of[k]] if k in second_of else None
                items.append((w, w, so, si, getattr(layer, 'groups', 1)))
                if so is not None:
                    vecs = [layer.bias]
                    kb = relations[first_of[k]].get_idxs()[2]
                    if kb is not None:
                        vecs += [getattr(graph[kb], 'fake_weight', None), getattr(graph[kb], 'fake_bias', None)]
                    for v in vecs:
                        if v is not None:
                            vb = bind(v)
                            items.append((vb, vb, so, None, 1))
            self._rebuild_items = items
            self.rebuild = None if use_torch_rebuild or not items else _RebuildPlan(items)

            # ---- this rank's components: scratch copies + the engine plan over them ----
            mine = [i for i, o in enumerate(self.owner) if o == self.rank]
            self.mine = [relations[i] for i in mine]
            self.session, self.snapshot, self._snap_items, self._sgraph_bind = None, None, [], []
            if mine:
                keys = [k for k in graph if any(k in relations[i].get_idxs()[:2] for i in mine)]
                bn_keys = [relations[i].get_idxs()[2] for i in mine if relations[i].get_idxs()[2] is not None]
                srcs = []
                for k in keys:
                    srcs.append(bind(graph[k].weight))
                    if k in first_of:
                        srcs.append(bind(graph[k].bias))
                for kb in bn_keys:
                    for name in ('fake_weight', 'fake_bias'):
                        if getattr(graph[kb], name, None) is not None:
                            srcs.append(bind(getattr(graph[kb], name)))
                sizes = [(t.numel() + 63) // 64 * 64 for t in srcs]              # 256-byte aligned carve-outs
                self.scratch = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
                clone, off = {}, 0
                for t, n in zip(srcs, sizes):
                    clone[t.data_ptr()] = self.scratch[off:off + t.numel()].view(t.shape)
                    off += n
                self._snap_items = [(t, clone[t.data_ptr()], None, None, 1) for t in srcs]
                self.snapshot = None if use_torch_rebuild else _RebuildPlan(self._snap_items)
                scale_slices = [self.flat[self.offsets[i]:self.offsets[i + 1]] for i in mine]
                if le_runner is not None:
                    sgraph = OrderedDict(graph)                                  # the same keys, owned modules replaced by copies
                    for k in keys + bn_keys:
                        sgraph[k] = copy.deepcopy(graph[k])
                    self._sgraph_bind = [(sgraph[k], graph[k]) for k in keys + bn_keys]
                    self.session = _RunnerSession(le_runner, sgraph, self.mine, targ_type, s_range, signed, eps, scale_slices)
                else:
                    index = {k: j for j, k in enumerate(keys)}
                    layers = [(clone[bind(graph[k].weight).data_ptr()],
                               clone[bind(graph[k].bias).data_ptr()] if k in first_of else None,
                               getattr(graph[k], 'groups', 1)) for k in keys]
                    rels = []
                    for i, sl in zip(mine, scale_slices):
                        a, b, kb = relations[i].get_idxs()
                        bn = graph[kb] if kb is not None else None
                        fw, fb = getattr(bn, 'fake_weight', None), getattr(bn, 'fake_bias', None)
                        rels.append((index[a], index[b], clone[bind(fw).data_ptr()] if fw is not None else None,
                                     clone[bind(fb).data_ptr()] if fb is not None else None, sl))
                    self.session = _EngineSession(layers, rels, s_range, signed, eps)

    # ---- the two table-driven launches (torch stand-ins in the CPU-only logic tests) ----
    def _run_snapshot(self):
        if self._sgraph_bind:                                # runner stand-in: refresh the copied modules instead
            for dst_m, src_m in self._sgraph_bind:
                dst_m.load_state_dict(src_m.state_dict())
        elif self._torch_rebuild:
            for it in self._snap_items:
                _torch_rebuild(*it)
        elif self.snapshot is not None:
            self.snapshot.run()

    def _run_rebuild(self):
        if self._torch_rebuild:
            for it in self._rebuild_items:
                _torch_rebuild(*it)
        elif self.rebuild is not None:
            self.rebuild.run()

    def run(self, max_sweeps=None, converge_thres=2e-7, converge_count=20, check=True):
        """Returns the number of sweeps.  ``max_sweeps=N`` pins the count (no exchange before the final all_gather);
        ``None`` keeps the reference's data-dependent loop with one 8-byte all_reduce per sweep.

        With a pinned count nothing on this path waits for the GPU: snapshot, sweeps, all_gather and rebuild are enqueued back
        to back and ``check`` (default) synchronises ONCE at the end to surface an abandoned in-launch wait of the sweeps --
        the rebuilt tensors are undefined then and the caller's weights must be reloaded (errors are never silent,
        dfq.py:126,276; the collective itself has the usual torch.distributed failure mode).  ``check=False`` leaves that
        to a later ``check()`` call (bench.py: the bias correction is enqueued right behind)."""
        if self._ran:
            raise RuntimeError('ShardedEqualizer.run() is single-use: the in-place rebuild has consumed the pristine tensors '
                               'and the accumulated scales; construct a new ShardedEqualizer for another pass')
        self._ran = True
        group, session = self.group, self.session
        with torch.no_grad():
            try:
                if session is not None:
                    self._run_snapshot()                     # owned tensors -> scratch (the real ones stay pristine)
                    session.start()
                if max_sweeps is not None:
                    if session is not None and max_sweeps > 0:
                        session.sweeps(max_sweeps)
                    sweeps = max_sweeps
                elif session is None or isinstance(session, _EngineSession):
                    sweeps = self._run_data_dependent(session, converge_thres, converge_count)
                else:
                    diff, count, sweeps = 10.0, 0, 0
                    while diff > converge_thres and count < converge_count:       # dfq.py:83
                        local = 0.0
                        if session is not None:
                            session.sweeps(1)
                            local = session.last_diff()
                        t = torch.tensor([local], dtype=torch.float64, device=self.comm_dev)
                        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                        diff_tmp = float(t.item())
                        if abs(diff - diff_tmp) > 1e-9:                           # dfq.py:110-115
                            count, diff = 0, diff_tmp
                        else:
                            count += 1
                        sweeps += 1
                if session is not None and self._sgraph_bind:
                    session.finish()                         # (runner stand-in: copies its scales into the exchange buffer)
            except Exception:
                if session is not None:
                    session.close()
                    self.session = None
                raise
            # ---- exchange: ONE all_gather of the cumulative scale vectors (RCCL over xGMI when the group is 'nccl') ----
            if self.comm_dev == self.dev:
                dist.all_gather_into_tensor(self.gathered, self.flat, group=group)
            else:
                g = torch.empty(self.gathered.shape, dtype=torch.float32, device=self.comm_dev)
                dist.all_gather_into_tensor(g, self.flat.to(self.comm_dev), group=group)
                self.gathered.copy_(g)
            # ---- every rank rebuilds every paired tensor from its pristine value: identical launch, identical inputs ----
            self._run_rebuild()
            if self.stage._shadow:
                self.check()                                 # CPU-resident tensors: the write-back below reads the results
            self.stage.writeback()
            S = self.gathered[self._sel]                    # the owners' segments, relation after relation (one gather)
            for i, rr in enumerate(self.relations):          # Relation.set_scale_vec, cumulative (relation.py:20-24)
                s = self.stage.out_like(self.graph[rr.get_idxs()[0]].weight, S[self.offsets[i]:self.offsets[i + 1]])
                rr.S = s if rr.S is None else rr.S.to(s.device) * s
        if check:
            self.check()
        return sweeps

    #: sweeps per chunk of the data-dependent mode: one collective and one host read per chunk (DFQ_SHARD_CHUNK)
    CHUNK = 12

    def _run_data_dependent(self, session, converge_thres, converge_count):
        """The reference's own stopping rule (dfq.py:83-115) over the ranks' summed mean|dW| WITHOUT a host round trip per sweep.
        A rank runs a chunk of sweeps with its plan's exit test disabled; the plan leaves diff_tmp of every sweep in a device
        log; ONE all_reduce per chunk sums the ranks' logs; the (diff, count) state machine runs over the chunk's sums on the
        device (dfq_le_shared_verdict) and the host reads four numbers per chunk.  A loop that stops INSIDE a chunk has run a few
        sweeps too many on this rank's SCRATCH copies (all a rank keeps of them are the cumulative scales): scratch tensors and
        scales go back to the chunk's start (two device copies taken there) and exactly the sweeps that happen are run again.
        Every rank sees the same sums, draws the same verdicts and enqueues the same collectives."""
        import os
        chunk = max(1, int(os.environ.get('DFQ_SHARD_CHUNK', self.CHUNK)))
        dev, cdev, group = self.dev, self.comm_dev, self.group
        log_cap = 64 * chunk
        log = torch.zeros(log_cap, dtype=torch.float64, device=dev)
        ext = torch.tensor([10.0, 0.0, 0.0, 0.0], dtype=torch.float64, device=dev)      # dfq.py:81-82
        keep_scratch = torch.empty_like(self.scratch) if session is not None else None
        keep_flat = torch.empty_like(self.flat)
        if session is not None:
            session.set_log(log)
        plan = session.plan._plan if session is not None else None
        engine_sweep, total = 0, 0            # sweeps since the plan's last restart / since the start of the pass
        try:
            while True:
                if engine_sweep + chunk > log_cap:                # (a loop of thousands of sweeps: the log starts over)
                    session is not None and session.start()
                    engine_sweep = 0
                if session is not None:
                    keep_scratch.copy_(self.scratch)
                keep_flat.copy_(self.flat)
                if session is not None:
                    session.sweeps(chunk)
                piece = log[engine_sweep:engine_sweep + chunk]
                if cdev == dev:
                    dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=group)
                else:
                    t = piece.to(cdev)
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                    piece.copy_(t)
                _ffi.check(_ffi.lib().dfq_le_shared_verdict(plan, piece.data_ptr(), chunk, ext.data_ptr(), float(converge_thres),
                                                            int(converge_count), -1, _ffi.stream_arg()))
                state = ext.tolist()                              # the ONE host read of the chunk
                engine_sweep += chunk
                if state[3] != 0.0:
                    n_keep = int(state[2]) - total
                    total = int(state[2])
                    if n_keep < chunk:
                        # the loop stopped inside the chunk: back to the chunk's start, then exactly the sweeps that happen
                        if session is not None:
                            self.scratch.copy_(keep_scratch)
                        self.flat.copy_(keep_flat)
                        if session is not None:
                            session.set_log(None)
                            session.start()                       # (statistics of the restored tensors; the loop state starts over)
                            if n_keep > 0:
                                session.sweeps(n_keep)
                    return total
                total += chunk
        finally:
            if session is not None and session.plan._plan:
                session.set_log(None)

    def check(self):
        """Synchronise and raise if a workgroup of this rank's sweeps abandoned a wait (once; closes the sweep plan)."""
        session, self.session = self.session, None
        if session is not None:
            try:
                if not self._sgraph_bind:
                    session.finish()
            finally:
                session.close()

    def close(self):
        for p in (self.rebuild, self.snapshot):
            if p is not None:
                p.close()
        if self.session is not None:
            self.session.close()
            self.session = None


def sharded_cross_layer_equalization(graph, relations, targ_type, group=None, s_range=(1e-8, 1e8),
                                     converge_thres=2e-7, converge_count=20, signed=False, eps=0,
                                     max_sweeps=None, le_runner=None, use_torch_rebuild=False):
    """Equalise ``graph`` in place on every rank of ``group``; returns the number of sweeps.  Every rank ends with
    the same bits in every tensor (see the module docstring).

    ``le_runner(graph, relations, targ_type, max_sweeps=..., **kw) -> dict`` and ``use_torch_rebuild`` replace the
    HIP engine's sweeps / batched rebuild launch by CPU stand-ins in the partition / exchange logic tests.

    ``max_sweeps=N`` pins the sweep count (no exchange before the final all_gather -- the mode for networks
    whose reference loop does not terminate, SURVEY 7.3 item 4, and the one bench.py times for config 4);
    ``max_sweeps=None`` keeps the reference's data-dependent loop: one 8-byte all_reduce per sweep.
    """
    eq = ShardedEqualizer(graph, relations, targ_type, group=group, s_range=s_range, signed=signed, eps=eps,
                          le_runner=le_runner, use_torch_rebuild=use_torch_rebuild)
    try:
        return eq.run(max_sweeps=max_sweeps, converge_thres=converge_thres, converge_count=converge_count)
    finally:
        eq.close()
