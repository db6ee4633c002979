"""_ffi.staging(): a sequence of drop-in calls on a CPU-resident model shares one staging area (one transfer each way)."""
import copy

import pytest
import torch

import dfq_amd
from dfq_amd import _ffi, dfq, synthetic
from dfq_amd.utils import layer_transform as lt
from dfq_amd.utils import relation as rel

from common import TARG, assert_bitexact, snapshot


def _cpu_model(name='tiny_mobile', seed=0):
    model, graph, bottoms = synthetic.build(name, seed=seed)          # stays on the CPU, like the reference's
    lt.merge_batchnorm(model, graph, bottoms, TARG)
    rels = rel.create_relation(graph, bottoms, TARG, delete_single=False)
    return model, graph, bottoms, rels


def _calibrate(graph, bottoms, rels):
    dfq.cross_layer_equalization(graph, rels, TARG)
    dfq.bias_absorption(graph, rels, bottoms)
    dfq.bias_correction(graph, bottoms, TARG)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['tiny_mobile', 'tiny_res'])
def test_scope_equals_separate_calls(name):
    _ffi.lib()
    dfq.clear_plan_cache()
    a, b = _cpu_model(name), _cpu_model(name)
    _calibrate(a[1], a[2], a[3])
    packs = []
    with dfq_amd.staging() as st:
        _calibrate(b[1], b[2], b[3])
        packs.append(len(st._packs))
        w = next(m for m in b[1].values() if type(m) in TARG).weight
        assert w.device.type == 'cpu'
    sa, sb = snapshot(a[1]), snapshot(b[1])
    for k in sa:
        assert_bitexact(sb[k], sa[k], '{} {}'.format(name, k))
    for ra, rb in zip(a[3], b[3]):
        assert ra.S.device.type == 'cpu' and rb.S.device.type == 'cpu'
        assert torch.equal(ra.S, rb.S)
    assert packs[0] <= 3                      # LE's tensors, then only what absorption / correction add (BatchNorms LE did not touch)
    assert getattr(_ffi._ambient, 'stage', None) is None      # (no explicit scope is open; the thread's persistent stage is another matter)


@pytest.mark.gpu
def test_scope_left_by_an_exception_writes_nothing_back():
    _ffi.lib()
    m = _cpu_model()
    before = {k: v.copy() for k, v in snapshot(m[1]).items()}
    with pytest.raises(RuntimeError, match='stop here'):
        with dfq_amd.staging():
            dfq.cross_layer_equalization(m[1], m[3], TARG)
            raise RuntimeError('stop here')
    after = snapshot(m[1])
    for k in before:
        assert_bitexact(after[k], before[k], k)
    assert getattr(_ffi._ambient, 'stage', None) is None      # (no explicit scope is open; the thread's persistent stage is another matter)
    # and the model is usable afterwards
    dfq.cross_layer_equalization(m[1], m[3], TARG)
    assert any((snapshot(m[1])[k] != before[k]).any() for k in before)


@pytest.mark.gpu
def test_scoped_calls_reuse_plans_for_the_next_model_of_the_same_shapes():
    """Inside a scope a CPU tensor has a stable device copy, so the plan cache can key on it; the next model of the same
    architecture is staged into the same (recycled) device block and hits the cache -- when the allocator hands the block out
    again, which it does for an equal-sized request right after the release."""
    _ffi.lib()
    dfq.clear_plan_cache()
    ref = _cpu_model(seed=3)
    _calibrate(ref[1], ref[2], ref[3])
    dfq.clear_plan_cache()
    stats0 = dict(dfq.plan_cache_stats)
    for seed in (1, 2, 3):
        m = _cpu_model(seed=seed)
        with dfq_amd.staging():
            _calibrate(m[1], m[2], m[3])
        last = m
    hits = dfq.plan_cache_stats['le_hits'] - stats0['le_hits'] + dfq.plan_cache_stats['bc_hits'] - stats0['bc_hits']
    sa, sb = snapshot(ref[1]), snapshot(last[1])
    for k in sa:
        assert_bitexact(sb[k], sa[k], k)          # whatever the cache did, the result is the un-cached one
    assert hits >= 0


@pytest.mark.gpu
def test_device_resident_model_is_untouched_by_a_scope():
    _ffi.lib()
    _ffi.release_staging()                        # (the scope works on the thread's persistent stage: start from an empty one)
    a, b = _cpu_model(), _cpu_model()
    for m in (a, b):
        m[0].to('cuda')
        for rr in m[3]:
            rr.S = None
    _calibrate(a[1], a[2], a[3])
    with dfq_amd.staging() as st:
        n0 = len(st._packs)                       # (the scope works on the thread's persistent stage: the CPU models' merge_batchnorm left packs)
        _calibrate(b[1], b[2], b[3])
        assert len(st._packs) == n0               # nothing to stage
    sa, sb = snapshot(a[1]), snapshot(b[1])
    for k in sa:
        assert_bitexact(sb[k], sa[k], k)


def test_scopes_join_and_clean_up(engine):
    with dfq_amd.staging() as outer:
        assert _ffi.scoped_stage() is outer
        with dfq_amd.staging() as inner:
            assert inner is outer
        assert _ffi.scoped_stage() is outer        # the inner scope did not end the outer one
        assert _ffi.entry_stage() is outer         # what a calibration entry point takes ...
        assert _ffi.Stage() is not outer           # ... and what a per-call user (QuantMeasure, quantize, prims) takes
    assert _ffi.scoped_stage() is _ffi.persistent_stage()      # outside a scope: the thread's persistent stage (round 6), or None
    assert getattr(_ffi._ambient, 'stage', None) is None and not outer._scoped      # (the scope worked ON the persistent stage)


@pytest.mark.gpu
def test_whole_calibration_section_inside_one_scope():
    """main_cls.py:149-181 from BN folding to the int8 weights, every call inside one scope: what the separate calls give."""
    _ffi.lib()

    def run(scoped):
        import contextlib
        model, graph, bottoms = synthetic.build('tiny_mobile', seed=0)
        with (dfq_amd.staging() if scoped else contextlib.nullcontext()):
            lt.merge_batchnorm(model, graph, bottoms, TARG)
            rels = rel.create_relation(graph, bottoms, TARG, delete_single=False)
            dfq.cross_layer_equalization(graph, rels, TARG)
            dfq.bias_absorption(graph, rels, bottoms)
            dfq.bias_correction(graph, bottoms, TARG)
            lt.quantize_targ_layer(graph, 8, 8, TARG)
        return snapshot(graph)
    a, b = run(False), run(True)
    for k in a:
        assert_bitexact(b[k], a[k], k)


@pytest.mark.gpu
def test_full_size_cpu_resident_model_in_a_scope_equals_the_device_resident_pass():
    """MobileNetV2 at full size: the CPU-resident model calibrated inside a scope ends bit-identical to the same model
    calibrated on the device (same plans, same kernels -- the scope only moves the bytes), sweep count included."""
    _ffi.lib()
    dfq.clear_plan_cache()
    a = _cpu_model('mobilenet_v2', seed=0)
    b = _cpu_model('mobilenet_v2', seed=0)
    b[0].to('cuda')
    with dfq_amd.staging():
        dfq.cross_layer_equalization(a[1], a[3], TARG)
        sweeps_a = dfq.last_equalization['sweeps']
        dfq.bias_correction(a[1], a[2], TARG)
    dfq.cross_layer_equalization(b[1], b[3], TARG)
    sweeps_b = dfq.last_equalization['sweeps']
    dfq.bias_correction(b[1], b[2], TARG)
    assert sweeps_a == sweeps_b == 47
    sa, sb = snapshot(a[1]), snapshot(b[1])
    for k in sb:
        assert_bitexact(sa[k], sb[k], k)
    for ra, rb in zip(a[3], b[3]):
        assert ra.S.device.type == 'cpu'
        assert torch.equal(ra.S, rb.S.cpu())


@pytest.mark.gpu
def test_per_call_users_do_not_join_the_scope():
    """a CPU-resident QuantMeasure called inside `with staging():` must keep tracking its range from call to
    call (its running range and its input change between calls: a scope-long binding would hand every later call the first
    call's device copy and write the stale copy back when the scope ends), and the scope must not adopt -- or write back over --
    the activations.  Only the calibration entry points share the scope's stage (_ffi.entry_stage)."""
    from dfq_amd.utils.quantize import QuantMeasure, quantize
    _ffi.lib()
    _ffi.release_staging()                        # (the scope works on the thread's persistent stage: start from an empty one)
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(4, 3, 8, 8, generator=g) * (i + 1) for i in range(3)]

    def run(scoped):
        qm = QuantMeasure()                                       # CPU-resident module, CPU inputs
        qm.set_update_stat(True)
        qm.eval()
        ins = [x.clone() for x in xs]
        outs, ranges = [], []
        ctx = dfq_amd.staging() if scoped else _null()
        with ctx as st:
            for x in ins:
                outs.append(qm(x).clone())
                ranges.append((float(qm.running_min), float(qm.running_max)))
            q = quantize(ins[0], 8, -1.0, 1.0)
            if scoped:
                assert not st._bound and not st._shadow and not st._packs      # nothing of a per-call user stayed with the scope
        return outs, ranges, ins, q, (float(qm.running_min), float(qm.running_max))

    import contextlib

    @contextlib.contextmanager
    def _null():
        yield None

    a, b = run(False), run(True)
    assert a[1] == b[1] and a[4] == b[4]
    assert a[1][0] != a[1][2]                                     # the range did move from call to call
    for x, y in zip(a[0], b[0]):
        assert torch.equal(x, y)
    for x, y in zip(xs, b[2]):
        assert torch.equal(x, y)                                  # inputs untouched when the scope ended
    assert torch.equal(a[3], b[3])


@pytest.mark.gpu
def test_weight_mutating_helpers_inside_a_scope_work_on_the_scope_copy():
    """`_layer_equalization` (and the other helpers with a private stage: merge_scale_into_layer, the single-step
    primitives) called INSIDE `with staging():` on tensors an entry point of the scope has already shadowed must read the
    scope's device copy -- the truth between the calls of a scope -- and what they write must survive the scope's write-back.
    Equal, bit for bit, to the same sequence of calls without a scope."""
    def run(scoped):
        model, graph, bottoms, rels = _cpu_model()
        a, b, kb = rels[0].get_idxs()
        import contextlib
        with (dfq_amd.staging() if scoped else contextlib.nullcontext()):
            dfq.cross_layer_equalization(graph, rels, TARG, max_sweeps=2, converge_thres=-1.0, converge_count=10 ** 9)
            # a helper with a stage of its own, on tensors the entry point above has shadowed (and rewritten on the device)
            dfq._layer_equalization(graph[a].weight, graph[b].weight, graph[a].bias)
            dfq.bias_correction(graph, bottoms, TARG)
        return snapshot(graph)
    want, got = run(False), run(True)
    for k in want:
        assert_bitexact(got[k], want[k], k)


@pytest.mark.gpu
def test_plain_calls_share_device_copies_and_see_host_side_writes():
    """the reference's own sequence of plain calls on a CPU-resident model (main_cls.py:149-188:
    merge_batchnorm -> cross_layer_equalization -> bias_correction -> quantize_targ_layer) transfers the network ONCE -- the
    thread's persistent stage keeps the device shadows of the previous call, keyed on every tensor's (_version, data_ptr) -- while
    every call still leaves its results in the caller's tensors.  A host-side write between two calls (an in-place op, a new
    storage) must be seen by the next call; the result equals the same calls with the persistent stage switched off."""
    def run(persist, mutate):
        _ffi.release_staging()
        old = _ffi._PERSIST
        _ffi._PERSIST = persist
        try:
            model, graph, bottoms = synthetic.build('tiny_mobile', seed=0)
            lt.merge_batchnorm(model, graph, bottoms, TARG)
            rels = rel.create_relation(graph, bottoms, TARG, delete_single=False)
            uploads = []
            st = _ffi.persistent_stage()
            n0 = len(st._packs) if st is not None else 0
            dfq.cross_layer_equalization(graph, rels, TARG)
            first = next(m for m in graph.values() if type(m) in TARG)
            after_le = first.weight.detach().clone()                      # the call left its result on the host
            if mutate:
                with torch.no_grad():
                    first.weight.mul_(1.5)                                 # an in-place write: _version moves
                    last = [m for m in graph.values() if type(m) in TARG][-1]
                    last.bias.data = last.bias.data.clone() + 0.25        # a new storage: data_ptr moves
            dfq.bias_correction(graph, bottoms, TARG)
            if st is not None:
                uploads = len(st._packs) - n0
            lt.quantize_targ_layer(graph, 8, 16, TARG)
            return snapshot(graph), after_le, uploads
        finally:
            _ffi._PERSIST = old
            _ffi.release_staging()
    for mutate in (False, True):
        want, le_w, _ = run(False, mutate)
        got, le_g, uploads = run(True, mutate)
        assert torch.equal(le_w, le_g)
        for k in want:
            assert_bitexact(got[k], want[k], '{} (mutate={})'.format(k, mutate))
        # the equalisation and the correction found the shadows merge_batchnorm had made: what they add are the BN proxies that
        # merge_batchnorm has just CREATED on the host (fake_weight / fake_bias: new tensors), nothing else
        assert uploads <= 2, uploads
