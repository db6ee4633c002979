"""N > 1 path of row (e): one network sharded over ranks (dfq_amd/sharded.py), world_size 2 over gloo
on CPU.  Two flavours: (1) injected stand-ins (numpy oracle per rank, torch ops for the rebuild) isolate the
partition / all_gather exchange of the scale vectors / rebuild logic; (2) the product code path -- the engine's
sweeps on every rank and the engine's row/column rescale kernels for the foreign layers -- with the kernels
running on the CPU emulation.  Both against the single-process oracle result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from oracle import dfq_oracle as orc
from oracle import graphspec
from dfq_amd import sharded, synthetic
from dfq_amd.utils import relation as rel

from common import TARG, assert_bitexact, assert_close, load_inputs, net_fixture, npy


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _prepare(name, seed):
    model, graph, bottoms = synthetic.build(name, seed=seed)
    if name.startswith('tiny_'):               # the tiny nets start from the fixture's inputs; full-size ones from their seed
        load_inputs(graph, net_fixture(name, seed, ''), 'cpu')
    spec = graphspec.from_torch(graph, bottoms, TARG)
    orc.merge_batchnorm(spec)
    with torch.no_grad():                      # put the oracle's folded state into the modules
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


def _oracle_runner(graph, relations, targ_type, max_sweeps=None, **kw):
    """Stand-in for the engine: run the numpy oracle on the given relations and write the result back."""
    spec = graphspec.GraphSpec()
    bottoms = {k: [] for k in graph}
    for k in graph:
        m = graph[k]
        if type(m) in targ_type:
            n = graphspec.Node(k, 'targ')
            n.weight = npy(m.weight).copy()
            n.bias = npy(m.bias).copy() if m.bias is not None else None
        elif type(m) == nn.BatchNorm2d and hasattr(m, 'fake_weight'):
            n = graphspec.Node(k, 'bn')
            n.fake_weight = npy(m.fake_weight).copy()
            n.fake_bias = npy(m.fake_bias).copy()
        else:
            n = graphspec.Node(k, 'other')
        spec.add(n, bottoms[k])
    rels = [rr.get_idxs() for rr in relations]
    trace = []
    n_sw, S = orc.cross_layer_equalization(spec, rels, s_range=kw.get('s_range', (1e-8, 1e8)),
                                           converge_thres=kw.get('converge_thres', 2e-7),
                                           converge_count=kw.get('converge_count', 20), signed=kw.get('signed', False),
                                           eps=kw.get('eps', 0), max_sweeps=max_sweeps, trace=trace)
    with torch.no_grad():
        for k in graph:
            n = spec.nodes[k]
            if n.kind == 'targ':
                graph[k].weight.copy_(torch.from_numpy(n.weight))
                if n.bias is not None:
                    if graph[k].bias is None:
                        graph[k].bias = nn.Parameter(torch.zeros(n.bias.shape[0]), requires_grad=False)
                    graph[k].bias.copy_(torch.from_numpy(n.bias))
            elif n.kind == 'bn':
                graph[k].fake_weight.copy_(torch.from_numpy(n.fake_weight))
                graph[k].fake_bias.copy_(torch.from_numpy(n.fake_bias))
    for rr, s in zip(relations, S):
        rr.set_scale_vec(torch.from_numpy(s.copy()))
    return dict(sweeps=n_sw, last_diff_tmp=trace[-1] if trace else 0.0)


def _use_emulated_engine(emu_path):
    """In a spawned rank: route dfq_amd to the CPU emulation build of the kernels (as the `engine` fixture does)."""
    import ctypes
    from dfq_amd import _ffi
    _ffi._lib = _ffi.bind(ctypes.CDLL(emu_path))
    _ffi.target_device = lambda: torch.device('cpu')
    _ffi.current_stream = lambda: 0
    _ffi.synchronize = lambda: None


def _worker(rank, world, port, name, seed, max_sweeps, out_dir, emu_path, standin=False, finish=True):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        model, graph, bottoms, _ = _prepare(name, seed)
        rels = rel.create_relation(graph, bottoms, TARG)
        _use_emulated_engine(emu_path)
        if standin:           # numpy oracle per rank + torch ops for the rebuild: the partition / exchange logic alone
            sweeps = sharded.sharded_cross_layer_equalization(graph, rels, TARG, max_sweeps=max_sweeps,
                                                              le_runner=_oracle_runner, use_torch_rebuild=True)
        else:                 # the product code path: engine sweeps per rank, ONE batched engine rebuild launch
            sweeps = sharded.sharded_cross_layer_equalization(graph, rels, TARG, max_sweeps=max_sweeps)
        owner = sharded.assign_components(graph, rels, world)
        snap = {'sweeps': np.array(sweeps), 'owner': np.array(owner)}
        for i, k in enumerate(graph):
            m = graph[k]
            if type(m) in TARG:
                snap['L{}.w'.format(i)] = npy(m.weight)
                if m.bias is not None:
                    snap['L{}.b'.format(i)] = npy(m.bias)
            elif type(m) == nn.BatchNorm2d and hasattr(m, 'fake_weight'):
                snap['L{}.fw'.format(i)] = npy(m.fake_weight)
                snap['L{}.fb'.format(i)] = npy(m.fake_bias)
        for i, rr in enumerate(rels):
            snap['S{}'.format(i)] = npy(rr.get_scale_vec())
        if finish:            # the replicated tail of the pass: bias correction + int8 weights / 16-bit biases (engine kernels)
            from dfq_amd import dfq
            from dfq_amd.utils import layer_transform as lt
            dfq.bias_correction(graph, bottoms, TARG)
            for i, k in enumerate(graph):
                m = graph[k]
                if type(m) in TARG and m.bias is not None:
                    snap['C{}.b'.format(i)] = npy(m.bias)
                elif type(m) == nn.BatchNorm2d and hasattr(m, 'fake_weight'):
                    snap['C{}.fb'.format(i)] = npy(m.fake_bias)
            _, codes = lt.quantize_targ_layer(graph, 8, 16, TARG, return_codes=True)
            for j, k in enumerate(codes):
                snap['Q{}'.format(j)] = npy(codes[k])
        np.savez(os.path.join(out_dir, 'rank{}.npz'.format(rank)), **snap)
    finally:
        dist.destroy_process_group()


def _check_against_oracle(res_list, graph, spec, S_ref):
    """every tensor within 1e-5 of the sequential (single-process oracle) result, cumulative scales bit-identical"""
    keys = list(graph.keys())
    worst = 0.0
    for res in res_list:
        for i, k in enumerate(keys):
            n = spec.nodes[k]
            if n.kind == 'targ':
                worst = max(worst, assert_close(res['L{}.w'.format(i)], n.weight, 'w {}'.format(k)))
                if n.bias is not None and 'L{}.b'.format(i) in res:
                    assert_close(res['L{}.b'.format(i)], n.bias, 'b {}'.format(k))
            elif n.kind == 'bn' and n.fake_weight is not None:
                assert_close(res['L{}.fw'.format(i)], n.fake_weight, 'fw {}'.format(k))
                assert_close(res['L{}.fb'.format(i)], n.fake_bias, 'fb {}'.format(k))
        for i, s in enumerate(S_ref):
            assert_bitexact(res['S{}'.format(i)], s, 'S{}'.format(i))
    return worst


def _check_ranks_identical(r0, r1, need_tail):
    """All ranks end with the SAME BITS in every tensor -- equalised weights / biases / BN proxies, and then the replicated
    tail of the pass: corrected biases, BN proxies after bias correction, int8 weight codes."""
    kinds = set()
    assert sorted(r0.files) == sorted(r1.files)
    for k in r0.files:
        if k[0] in 'LCQS':
            assert_bitexact(r0[k], r1[k], 'rank 0 vs rank 1: ' + k)
            kinds.add(k[0])
    assert kinds >= (set('LCQS') if need_tail else set('LS'))


def _canonical(name, seed, S_ref):
    """diag(S_out) . W0 . diag(1/S_in) with numpy float32 operations, from the pristine tensors and the oracle's cumulative
    scales: what every rank must hold, bit for bit, for every world size."""
    model, graph, bottoms, spec = _prepare(name, seed)
    orels = orc.create_relation(spec)
    out = {}
    keys = list(graph.keys())
    for (a, b, bn), S in zip(orels, S_ref):
        S = S.astype(np.float32)
        na, nb = spec.nodes[a], spec.nodes[b]
        out.setdefault(a, na.weight.copy())
        out.setdefault(b, nb.weight.copy())
    for (a, b, bn), S in zip(orels, S_ref):      # rows first (fl(w0 * s_out)) ...
        S = S.astype(np.float32)
        out[a] = out[a] * S.reshape((-1,) + (1,) * (out[a].ndim - 1))
    for (a, b, bn), S in zip(orels, S_ref):      # ... then columns (fl(t / s_in))
        S = S.astype(np.float32)
        w = out[b]
        groups = getattr(graph[b], 'groups', 1)
        per_row = np.repeat(S.reshape(groups, -1), w.shape[0] // groups, axis=0)
        out[b] = w / per_row.reshape((w.shape[0], -1) + (1,) * (w.ndim - 2))
    return {'L{}.w'.format(keys.index(k)): v for k, v in out.items()}


@pytest.mark.parametrize('engine_kind', ['stand-in', 'emulated-engine'])
@pytest.mark.parametrize('name,seed,max_sweeps', [('tiny_mobile', 0, 5), ('tiny_mobile', 0, None), ('tiny_res', 0, 3)])
def test_sharded_equalization_two_ranks(tmp_path, emu_lib_path, name, seed, max_sweeps, engine_kind):
    world = 2
    standin = engine_kind == 'stand-in'
    mp.spawn(_worker, args=(world, _free_port(), name, seed, max_sweeps, str(tmp_path), emu_lib_path, standin),
             nprocs=world, join=True)
    # single-process result of the same algorithm
    model, graph, bottoms, spec = _prepare(name, seed)
    orels = orc.create_relation(spec)
    if max_sweeps is None:
        n_ref, S_ref = orc.cross_layer_equalization(spec, orels)
    else:       # the sharded pinned mode runs exactly max_sweeps sweeps
        n_ref, S_ref = orc.cross_layer_equalization(spec, orels, max_sweeps=max_sweeps, converge_thres=-1.0,
                                                    converge_count=10 ** 9)
    r0 = np.load(os.path.join(str(tmp_path), 'rank0.npz'))
    r1 = np.load(os.path.join(str(tmp_path), 'rank1.npz'))
    assert int(r0['sweeps']) == int(r1['sweeps']) == n_ref
    assert len(set(r0['owner'].tolist())) == 2, 'both ranks must own work'
    _check_against_oracle((r0, r1), graph, spec, S_ref)
    _check_ranks_identical(r0, r1, need_tail=True)
    # ... and it is THE canonical rebuild, whoever computed it (torch stand-in or the engine's batched launch)
    for k, v in _canonical(name, seed, S_ref).items():
        assert_bitexact(r0[k], v, 'canonical ' + k)


@pytest.mark.parametrize('chunk,cf_group', [('1', '4'), ('3', '4'), ('16', '4'), ('3', '1')])
def test_sharded_data_dependent_loop_in_chunks(tmp_path, emu_lib_path, monkeypatch, chunk, cf_group):
    """the reference's own stopping rule (dfq.py:83-115) over the ranks' summed mean|dW| without a host round
    trip per sweep -- chunks of sweeps, ONE all_reduce of a chunk's per-sweep sums, the verdicts drawn on the device
    (dfq_le_shared_verdict), scratch tensors and scales taken back to the chunk's start when the loop stops inside a chunk.
    Whatever the chunk size: the oracle's sweep count, bit-exact cumulative scales, bit-equal ranks."""
    monkeypatch.setenv('DFQ_SHARD_CHUNK', chunk)
    monkeypatch.setenv('DFQ_LE_CF_GROUP', cf_group)             # free-running segments of depth 4 / every layer on the general tiles (a small network's default)
    world, name, seed = 2, 'tiny_mobile', 0
    mp.spawn(_worker, args=(world, _free_port(), name, seed, None, str(tmp_path), emu_lib_path, False, False), nprocs=world, join=True)
    model, graph, bottoms, spec = _prepare(name, seed)
    n_ref, S_ref = orc.cross_layer_equalization(spec, orc.create_relation(spec))
    r0 = np.load(os.path.join(str(tmp_path), 'rank0.npz'))
    r1 = np.load(os.path.join(str(tmp_path), 'rank1.npz'))
    assert int(r0['sweeps']) == int(r1['sweeps']) == n_ref
    assert n_ref % 16 != 0                                       # (the loop does stop inside a chunk of 3 or 16)
    _check_against_oracle((r0, r1), graph, spec, S_ref)
    _check_ranks_identical(r0, r1, need_tail=False)


def test_sharded_result_does_not_depend_on_world_size(tmp_path, emu_lib_path):
    """One rank and two ranks end with the same bits (the rebuild is a function of W0 and the cumulative scales only)."""
    outs = []
    for world in (1, 2):
        d = tmp_path / 'w{}'.format(world)
        d.mkdir()
        mp.spawn(_worker, args=(world, _free_port(), 'tiny_mobile', 0, 5, str(d), emu_lib_path), nprocs=world, join=True)
        outs.append(np.load(os.path.join(str(d), 'rank0.npz')))
    for k in outs[0].files:
        if k[0] in 'LCQS':
            assert_bitexact(outs[0][k], outs[1][k], 'world 1 vs world 2: ' + k)


def test_sharded_deeplab_two_ranks_product_path(tmp_path, emu_lib_path):
    """BASELINE.json config 4: DeepLab-v3+ (MobileNetV2 backbone, 61 convs, 35 relations) sharded over ranks, pinned to
    4 sweeps here (the CPU emulation is slow; 60 on the GPU: test_sharded_network_over_rccl_all_ranks, bench.py -- the
    reference's loop does not terminate on this network) -- the product code path
    (per-rank engine plan over scratch copies, ONE all_gather of the cumulative scale vectors, ONE batched rebuild launch on
    every rank, then the replicated bias correction and int8 quantisation) on the CPU emulation of the kernels, world size 2
    over gloo, against the single-process oracle: cumulative scales bit-exact, every tensor within 1e-5, and the two ranks
    bit-identical in every tensor, every corrected bias and every int8 code."""
    world, name, seed, sweeps = 2, 'deeplab_mnv2', 0, 4
    mp.spawn(_worker, args=(world, _free_port(), name, seed, sweeps, str(tmp_path), emu_lib_path), nprocs=world, join=True)
    model, graph, bottoms, spec = _prepare(name, seed)
    orels = orc.create_relation(spec)
    assert len(orels) == 35          # the reference's DeepLab graph (tests/golden/graph_deeplab_mnv2_relu.json, full_deeplab_mnv2_s0.npz)
    n_ref, S_ref = orc.cross_layer_equalization(spec, orels, max_sweeps=sweeps, converge_thres=-1.0, converge_count=10 ** 9)
    r0 = np.load(os.path.join(str(tmp_path), 'rank0.npz'))
    r1 = np.load(os.path.join(str(tmp_path), 'rank1.npz'))
    assert int(r0['sweeps']) == int(r1['sweeps']) == n_ref == sweeps
    assert sorted(set(r0['owner'].tolist())) == [0, 1]
    _check_against_oracle((r0, r1), graph, spec, S_ref)
    _check_ranks_identical(r0, r1, need_tail=True)
    for k, v in _canonical(name, seed, S_ref).items():
        assert_bitexact(r0[k], v, 'canonical ' + k)


def test_components_and_assignment():
    model, graph, bottoms, _ = _prepare('tiny_mobile', 0)
    rels = rel.create_relation(graph, bottoms, TARG)
    comps = sharded.relation_components(rels)
    assert sorted(i for c in comps for i in c) == list(range(len(rels)))
    layers_of = [set(k for i in c for k in rels[i].get_idxs()[:2]) for c in comps]
    for a in range(len(comps)):
        for b in range(a + 1, len(comps)):
            assert not (layers_of[a] & layers_of[b]), 'components must not share layers'
    for world in (1, 2, 3, 8):
        owner = sharded.assign_components(graph, rels, world)
        assert len(owner) == len(rels) and all(0 <= o < world for o in owner)
        for c in comps:
            assert len({owner[i] for i in c}) == 1, 'a component stays on one rank'


def test_sharded_equalizer_is_single_use(emu_lib_path):
    """a second run() would rebuild from already-rescaled tensors with scales accumulated over both runs
    (W0.S1.(S1.S2)).  The object is single-use and says so."""
    import ctypes
    from dfq_amd import _ffi
    saved = (_ffi._lib, _ffi.target_device, _ffi.current_stream, _ffi.synchronize)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('gloo', rank=0, world_size=1)
    try:
        _use_emulated_engine(emu_lib_path)
        model, graph, bottoms, spec = _prepare('tiny_mobile', 0)
        rels = rel.create_relation(graph, bottoms, TARG)
        eq = sharded.ShardedEqualizer(graph, rels, TARG)
        try:
            assert eq.run(max_sweeps=3) == 3
            first = {k: npy(graph[k].weight).copy() for k in graph if type(graph[k]) in TARG}
            with pytest.raises(RuntimeError, match='single-use'):
                eq.run(max_sweeps=3)
            for k, v in first.items():                       # the refused call touched nothing
                assert_bitexact(npy(graph[k].weight), v, k)
        finally:
            eq.close()
        orels = orc.create_relation(spec)
        _, S_ref = orc.cross_layer_equalization(spec, orels, max_sweeps=3, converge_thres=-1.0, converge_count=10 ** 9)
        for rr, s in zip(rels, S_ref):
            assert_bitexact(npy(rr.get_scale_vec()), s, 'S')
    finally:
        dist.destroy_process_group()
        _ffi._lib, _ffi.target_device, _ffi.current_stream, _ffi.synchronize = saved


@pytest.mark.gpu
def test_sharded_path_over_rccl_single_rank():
    """The RCCL ('nccl') branch of the sharded path on a real GPU: world size 1 (the GPU box has one device) -- the
    collectives, the device-side scale exchange, the engine's per-rank sweeps on scratch copies and the batched rebuild
    launch run for real.  Cumulative scales: bit-identical to the plain single-GPU equalisation; tensors: within 1e-5 of
    it and bit-identical to the canonical rebuild diag(S_out) . W0 . diag(1/S_in) computed with numpy -- the same bits a
    rank of a larger group would hold (tests above)."""
    from dfq_amd import dfq
    from dfq_amd.utils import layer_transform as lt
    from common import snapshot
    dev = torch.device('cuda', 0)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        for name, sweeps_pin in (('tiny_mobile', 5), ('deeplab_mnv2', 60)):
            out = []
            for use_sharded in (True, False):
                model, graph, bottoms = synthetic.build(name, seed=0)
                model.to(dev)
                lt.merge_batchnorm(model, graph, bottoms, TARG)
                rels = rel.create_relation(graph, bottoms, TARG)
                w0 = snapshot(graph)
                if use_sharded:
                    sweeps = sharded.sharded_cross_layer_equalization(graph, rels, TARG, max_sweeps=sweeps_pin)
                else:
                    dfq.cross_layer_equalization(graph, rels, TARG, max_sweeps=sweeps_pin, converge_thres=-1.0,
                                                 converge_count=10 ** 9)
                    sweeps = dfq.last_equalization['sweeps']
                out.append((sweeps, snapshot(graph), [npy(r.get_scale_vec()) for r in rels]))
            assert out[0][0] == out[1][0] == sweeps_pin
            for a, b in zip(out[0][2], out[1][2]):
                assert_bitexact(a, b, 'S')
            for k in out[1][1]:
                assert_close(out[0][1][k], out[1][1][k], k)
            # canonical rebuild from the pristine tensors and the cumulative scales (numpy float32: one rounding per operation)
            keys = list(graph.keys())
            want = {k: v.copy() for k, v in w0.items()}
            for rr, S in zip(rels, out[0][2]):
                a, b, bn = rr.get_idxs()
                ia = keys.index(a)
                want['L{}.w'.format(ia)] = want['L{}.w'.format(ia)] * S.reshape((-1,) + (1,) * (want['L{}.w'.format(ia)].ndim - 1))
                want['L{}.b'.format(ia)] = want.get('L{}.b'.format(ia), np.zeros_like(S)) * S
                if bn is not None:
                    ib = keys.index(bn)
                    want['L{}.fw'.format(ib)] = want['L{}.fw'.format(ib)] * S
                    want['L{}.fb'.format(ib)] = want['L{}.fb'.format(ib)] * S
            for rr, S in zip(rels, out[0][2]):
                a, b, bn = rr.get_idxs()
                ib = keys.index(b)
                w = want['L{}.w'.format(ib)]
                groups = getattr(graph[b], 'groups', 1)
                per_row = np.repeat(S.reshape(groups, -1), w.shape[0] // groups, axis=0)
                want['L{}.w'.format(ib)] = w / per_row.reshape((w.shape[0], -1) + (1,) * (w.ndim - 2))
            for k in want:
                assert_bitexact(out[0][1][k], want[k], 'canonical ' + k)
    finally:
        dist.destroy_process_group()


def _rccl_worker(rank, world, port, name, seed, sweeps, out_dir):
    """One rank of the REAL multi-GPU path: its own device, RCCL ('nccl') process group, product library."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    from dfq_amd import dfq
    from dfq_amd.utils import layer_transform as lt
    dev = torch.device('cuda', rank)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    try:
        model, graph, bottoms = synthetic.build(name, seed=seed)
        model.to(dev)
        lt.merge_batchnorm(model, graph, bottoms, TARG)
        rels = rel.create_relation(graph, bottoms, TARG)
        n = sharded.sharded_cross_layer_equalization(graph, rels, TARG, max_sweeps=sweeps)
        snap = {'sweeps': np.array(n), 'owner': np.array(sharded.assign_components(graph, rels, world))}
        for i, k in enumerate(graph):
            m = graph[k]
            if type(m) in TARG:
                snap['L{}.w'.format(i)] = npy(m.weight)
                if m.bias is not None:
                    snap['L{}.b'.format(i)] = npy(m.bias)
            elif type(m) == nn.BatchNorm2d and hasattr(m, 'fake_weight'):
                snap['L{}.fw'.format(i)] = npy(m.fake_weight)
                snap['L{}.fb'.format(i)] = npy(m.fake_bias)
        for i, rr in enumerate(rels):
            snap['S{}'.format(i)] = npy(rr.get_scale_vec())
        dfq.bias_correction(graph, bottoms, TARG)
        for i, k in enumerate(graph):
            m = graph[k]
            if type(m) in TARG and m.bias is not None:
                snap['C{}.b'.format(i)] = npy(m.bias)
            elif type(m) == nn.BatchNorm2d and hasattr(m, 'fake_weight'):
                snap['C{}.fb'.format(i)] = npy(m.fake_bias)
        _, codes = lt.quantize_targ_layer(graph, 8, 16, TARG, return_codes=True)
        for j, k in enumerate(codes):
            snap['Q{}'.format(j)] = npy(codes[k])
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, 'rank{}.npz'.format(rank)), **snap)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize('name,sweeps,n_rel', [('deeplab_mnv2', 60, 35), ('mobilenet_v2', 47, 37), ('mobilenet_v2', None, 37)])
def test_sharded_network_over_rccl_all_ranks(tmp_path, name, sweeps, n_rel):
    """BASELINE.json config 4 on the hardware it names, in the two configurations bench.py times (`sharded`): DeepLab (35
    relations, 60 pinned sweeps) and north_star's own graph, the 53-layer MobileNetV2 (37 relations in 16 components, 47 pinned
    sweeps), sharded over min(device_count, 8) ranks, one process per GPU, RCCL all_gather of the cumulative scale vectors over
    xGMI, then the replicated bias correction and int8 quantisation.  Every rank must hold the SAME BITS in every tensor,
    corrected bias and int8 code, and they must equal the world-size-1 result (spawned the same way).  On a 1-GPU box only the
    world-size-1 leg runs (the spawn / device binding / RCCL init code of the N > 1 leg is the same code)."""
    n_dev = torch.cuda.device_count()
    seed = 0
    worlds = [1] if n_dev < 2 else [1, min(n_dev, 8)]
    res = {}
    # (sweeps None: the reference's own stopping rule over the ranks' summed mean|dW| -- chunks of sweeps, one all_reduce per chunk,
    # verdicts on the device, sharded.py; the oracle's data-dependent loop says where it must stop)
    model, graph, bottoms, spec = _prepare(name, seed)
    orels = orc.create_relation(spec)
    assert len(orels) == n_rel
    if sweeps is None:
        pinned, (sweeps, S_ref) = None, orc.cross_layer_equalization(spec, orels)
    else:
        pinned = sweeps
        _, S_ref = orc.cross_layer_equalization(spec, orels, max_sweeps=sweeps, converge_thres=-1.0, converge_count=10 ** 9)
    for world in worlds:
        d = tmp_path / 'w{}'.format(world)
        d.mkdir()
        mp.spawn(_rccl_worker, args=(world, _free_port(), name, seed, pinned, str(d)), nprocs=world, join=True)
        res[world] = [np.load(os.path.join(str(d), 'rank{}.npz'.format(r))) for r in range(world)]
        assert all(int(r['sweeps']) == sweeps for r in res[world])
        for r in res[world][1:]:
            _check_ranks_identical(res[world][0], r, need_tail=True)
        if world > 1:
            assert len(set(res[world][0]['owner'].tolist())) > 1, 'more than one rank must own work'
    base = res[1][0]
    # the single-process oracle: cumulative scales bit-identical, tensors within 1e-5 (and the canonical rebuild bit for bit)
    for i, s in enumerate(S_ref):
        assert_bitexact(base['S{}'.format(i)], s, 'S{}'.format(i))
    for world in worlds[1:]:
        for k in base.files:
            if k[0] in 'LCQS':
                assert_bitexact(res[world][0][k], base[k], 'world {} vs world 1: {}'.format(world, k))
