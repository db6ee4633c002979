#!/usr/bin/env python
"""Headline benchmark: conv weights calibrated per second by the LE+BC pass over synthetic MobileNetV2 graphs
(BASELINE.json configs[1]: `--relu --equalize --correction`, 53 layers), plus one measured entry per other
BASELINE configuration in the same JSON line.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Headline (`value`).  A step = one complete pass of the hot path over a batch of `--batch` networks (distinct seeds)
whose folded weights are already resident in HBM: cross_layer_equalization until every network's reference
convergence test fires (device-side loop; the host enqueues the largest sweep count, every network leaves at its
own) followed by bias_correction.  Every step runs on its own replicas (in-place algorithm: a used replica is
"calibrated"; 288 GB of HBM hold hundreds), nothing is restored or skipped inside the timed region and the host
never synchronises between launches.  Multi-GPU: the unit that shards without any exchange is a network; rank r
calibrates its own replicas, `value` = weights calibrated by all ranks / max-over-ranks time -> "scaling": "weak".

Next to it, measured by the same run and reported in the same line:
  latency   one single-network pass with nothing else in flight (what main_cls.py would feel), with its own
            roofline fraction;
  config.others   configs[2] ResNet-18, configs[3] DeepLab on one GPU -- through the reference's own data-dependent loop (46
            sweeps: on its 35-relation graph the reference terminates, tests/golden/fullconv_deeplab_mnv2_s0.npz) and
            through the 60 pinned sweeps SURVEY 8d names --, configs[4] the activation-range kernels at MobileNetV2's
            largest activation (12 B per element);
  sharded   a LIST: `north_star`'s own graph (the 53-layer MobileNetV2, 47 pinned sweeps) and configs[3] (DeepLab, 60
            pinned sweeps), each as ONE network whose relation components are partitioned over the
            ranks, pinned sweeps, ONE all_gather of the cumulative scale vectors (RCCL over xGMI), engine rebuild
            of every paired layer, replicated bias correction -> strong scaling (total work fixed); runs at every
            N (at N = 1 it is the degenerate one-rank group), so a driver run at N = 8 puts 8 ranks on the data path;
            `data_dependent_ms`: the same pass with the reference's stopping rule (one all_reduce per chunk of sweeps);
  roofline  dominant kernel le_level_kernel: algorithmic bytes per launch / HIP-event duration per launch;
  cpu_baseline   the numpy oracle (a vectorised CPU port) timed live on this host + `reference`: the UNMODIFIED reference's own
            CPU path timed on THIS box's host cores (oracle/time_ref.py drives the byte-compiled reference of oracle/_ref for
            two sweeps + one bias correction and scales to the sweeps of a full pass; the committed build-container figure
            is carried only when oracle/_ref is absent).
"""
from __future__ import annotations

import argparse
import contextlib
import copy
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch                      # noqa: E402
import torch.nn as nn             # noqa: E402

TARG = [nn.Conv2d, nn.Linear]
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
NETS = ['mobilenet_v2', 'resnet18', 'deeplab_mnv2', 'tiny_mobile', 'tiny_res']


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--net', default='mobilenet_v2', choices=NETS)
    ap.add_argument('--sweeps', type=int, default=0, help='pin the LE sweep count (0 = what the convergence test needs)')
    ap.add_argument('--cpu-seconds', type=float, default=10.0, help='CPU-baseline budget (0 disables)')
    ap.add_argument('--batch', type=int, default=64, help='networks calibrated together in one step (one batched plan).  64 since '
                    'late round 6 (32 before): a sweep of the batch pays ~27 us of launch boundaries and convergence launch '
                    'whatever the batch, so a larger batch is a larger share of bytes moved -- one box, full default run: 1.77e10 '
                    'weights/s at 32, 1.95e10 at 64, 2.01e10 at 128 (119 s of wall time; 70 s at 64); profiles/r06_experiments.txt 10')
    ap.add_argument('--streams', type=int, default=2, help='steps in flight per GPU (one HIP stream + host thread each); '
                    'kernels with in-launch waits are serialised across streams by the library, the others overlap')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--force-sweeps', action='store_true', help='tuning: run --sweeps sweeps regardless of convergence')
    ap.add_argument('--others', default='resnet18,deeplab_mnv2,deeplab_mnv2:60', help='other BASELINE configs measured on one GPU: '
                    'comma list of net[:pinned sweeps]; "" disables')
    ap.add_argument('--act-shape', default='64,96,112,112', help='config 5: activation tensor for the range kernels; "" disables')
    ap.add_argument('--sharded', default='mobilenet_v2:47,deeplab_mnv2:60', help='config 4: net:pinned sweeps for the sharded single-network '
                    'pass; "" disables')
    ap.add_argument('--sharded-steps', type=int, default=6)
    ap.add_argument('--distill', default='mobilenet_v2:8:64,3,224,224', help='config 5 end to end: net:batches:shape of '
                    'update_quant_range over the distilled batches; "" disables')
    ap.add_argument('--pcie', default='mobilenet_v2', help='CPU-resident model through the drop-in entry points; "" disables')
    ap.add_argument('--lazy-steps', type=int, default=6, help='timed steps of the opt-in lazy-scale formulation; 0 disables')
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------
# set-up helpers (untimed)
# ---------------------------------------------------------------------------------------------------
def prepare(net, seed, dev):
    """Random-init network -> device -> BN folded -> relation list (untimed set-up)."""
    from dfq_amd import synthetic
    from dfq_amd.utils import layer_transform as lt
    from dfq_amd.utils import relation as rel
    model, graph, bottoms = synthetic.build(net, seed=seed)
    model.to(dev)
    lt.merge_batchnorm(model, graph, bottoms, TARG)
    rels = rel.create_relation(graph, bottoms, TARG, delete_single=False)
    return model, graph, bottoms, rels


ARENA = os.environ.get('DFQ_BENCH_ARENA', '1') != '0'      # batches as one allocation (dfq_amd/arena.py); 0: tensors where torch put them


def make_unit(protos):
    """One unit of work = a batch of networks calibrated together: deep copies of the prototypes plus
    one LE plan and one BC plan over the whole batch (a batch of 1 is an ordinary single-network plan)."""
    from dfq_amd import arena, dfq
    nets = [copy.deepcopy(p) for p in protos]
    if len(nets) == 1:
        t0 = time.perf_counter()
        model, graph, bottoms, rels = nets[0]
        le = dfq.build_le_plan(graph, rels, TARG)
        bc, _ = dfq.build_bc_plan(graph, bottoms, TARG)
        return dict(nets=nets, le=le, bc=bc, plan_build_ms=(time.perf_counter() - t0) * 1e3)
    if ARENA:
        # the batch is put together ONCE as one allocation with a fixed stride per network (part of loading the models, like
        # the deep copy above); a plan is then one network's tables + a base address per network
        t0 = time.perf_counter()
        batch = arena.NetworkBatch([(g, b, r) for (_, g, b, r) in nets], TARG)
        t1 = time.perf_counter()
        le, bc = batch.le_plan(), batch.bc_plan()
        return dict(nets=nets, le=le, bc=bc, batch=batch, layout_ms=(t1 - t0) * 1e3, plan_build_ms=(time.perf_counter() - t1) * 1e3)
    t0 = time.perf_counter()
    le = dfq.build_le_plan_batch([(g, r) for (_, g, _, r) in nets], TARG)
    bc = dfq.build_bc_plan_batch([(g, b) for (_, g, b, _) in nets], TARG)
    return dict(nets=nets, le=le, bc=bc, plan_build_ms=(time.perf_counter() - t0) * 1e3)


def pass_bytes(le, bc, sweeps):
    """Algorithmic bytes of one LE(sweeps)+BC pass as executed: per sweep 8 B per element read and written +
    4 B per interior element only measured, less the stores the streaming engine defers (LEPlan.sweep_bytes); bootstrap 4 B
    per paired element; BC 8 B per weight (min/max read + the chain's
    read: the quant-error row sums are formed in registers)."""
    return sweeps * le.sweep_bytes + 4 * le.paired_elements + 8 * bc.weight_elements


def cpu_baseline(net, seed, budget_s):
    """The numpy oracle (port of dfq.py's LE + BC, vectorised over channels) on this host."""
    from dfq_amd import synthetic
    from oracle import dfq_oracle as orc
    from oracle import graphspec
    torch.set_num_threads(1)
    model, graph, bottoms = synthetic.build(net, seed=seed)
    spec0 = graphspec.from_torch(graph, bottoms, TARG)
    orc.merge_batchnorm(spec0)
    rels = orc.create_relation(spec0)
    n_w = spec0.n_weights()
    reps, spent, sweeps = 0, 0.0, 0
    while reps == 0 or (spent < budget_s and reps < 64):
        spec = spec0.clone()
        t0 = time.perf_counter()
        sweeps, _ = orc.cross_layer_equalization(spec, rels)
        orc.bias_correction(spec)
        spent += time.perf_counter() - t0
        reps += 1
    out = dict(value=n_w * reps / spent, unit='weights/s', cores=1, kind='port',
               sample='{} full LE({} sweeps)+BC passes of the numpy oracle over the same synthetic {} '
                      '({} weights), {:.1f} s of CPU time'.format(reps, sweeps, net, n_w, spent))
    out['reference'] = reference_cpu_record(net, full_sweeps=sweeps)
    return out, sweeps


def reference_cpu_record(net, full_sweeps=0, timed_sweeps=2):
    """The UNMODIFIED reference's CPU path (dfq.py:78-117 + :173-293) on the same synthetic network, timed on THIS box's
    host cores: oracle/time_ref.py (a subprocess, so the reference's top-level `utils` package stays out of this process)
    drives the byte-compiled reference of oracle/_ref (oracle/build_ref.py, built by __graft_entry__.build() where
    /root/reference exists; git-ignored, travels with the snapshot) for `timed_sweeps` sweeps + one bias correction and
    extrapolates to the `full_sweeps` of a whole pass ("time k sweeps ... report s/sweep").  Only when oracle/_ref
    is absent the committed build-container figure (profiles/r02_reference_cpu.json) is carried instead, labelled as such."""
    import subprocess
    script = os.path.join(ROOT, 'oracle', 'time_ref.py')
    if os.path.isfile(os.path.join(ROOT, 'oracle', '_ref', 'dfq.pyc')):
        try:
            res = subprocess.run([sys.executable, script, '--net', net, '--sweeps', str(timed_sweeps),
                                  '--full-sweeps', str(full_sweeps)], capture_output=True, text=True, timeout=600)
            rec = json.loads(res.stdout.strip().splitlines()[-1])
            if 'error' not in rec:
                rec['measured_by_this_run'] = True
                return rec
        except Exception as e:                                  # fall through to the committed figure, say why
            print('reference timing failed: {!r}'.format(e), file=sys.stderr)
    path = os.path.join(ROOT, 'profiles', 'r02_reference_cpu.json')
    try:
        rec = json.load(open(path)).get(net)
    except Exception:
        rec = None
    if not rec:
        return None
    return {'value': rec['value'], 'unit': rec['unit'], 'cores': rec['cores'], 'kind': 'reference',
            'sweeps': rec['sweeps'], 'equalization_s': rec['equalization_s'], 'bias_correction_s': rec['bias_correction_s'],
            'measured_by_this_run': False,
            'what': rec['what'], 'where': rec['where'] + ' -- committed figure (profiles/r02_reference_cpu.json), not measured by this run'}


def _pmc_traffic(net, batch):
    """HBM bytes per launch of le_level_kernel from the committed PMC summary of a run with exactly this batch
    (rocprofv3 --pmc cannot run inside this process), else null."""
    for name in ('r06_pmc_summary.json', 'r06_pmc_summary_batch32.json'):      # (older summaries describe a kernel that moved other bytes: not this round's traffic)
        path = os.path.join(ROOT, 'profiles', name)
        if net != 'mobilenet_v2' or not os.path.exists(path):
            continue
        try:
            summary = json.load(open(path))
            if int(summary['batch']) == int(batch):
                return summary['le_level_kernel']['traffic_bytes_per_launch'], 'profiles/' + name
        except Exception:
            pass
    return None, None


def _pmc_kernel_traffic(kernel, net, batch):
    """Mean HBM bytes per DISPATCH of `kernel` from the committed PMC summary of a run at this batch size: FETCH_SIZE (KB,
    doubled as MI355X_MICROARCH.md prescribes) + WRITE_SIZE (KB), separate --pmc passes of tools/pmc_unit.py; else null."""
    for name in ('r06_pmc_summary.json', 'r06_pmc_summary_batch32.json', 'r05_pmc_summary.json'):
        path = os.path.join(ROOT, 'profiles', name)
        if net != 'mobilenet_v2' or not os.path.exists(path):
            continue
        try:
            summary = json.load(open(path))
            if int(summary['batch']) != int(batch):
                continue
            key = next(k for k in summary['FETCH_SIZE']['per_kernel_mean_KB'] if kernel in k)
            fetch = 2.0 * 1024.0 * float(summary['FETCH_SIZE']['per_kernel_mean_KB'][key])
            write = 1024.0 * float(summary['WRITE_SIZE']['per_kernel_mean_KB'][key])
            return {'kernel': kernel, 'bytes_per_dispatch': fetch + write, 'fetch_bytes_corrected': fetch, 'write_bytes': write,
                    'dispatches_in_pmc_run': int(summary['FETCH_SIZE']['per_kernel_dispatches'][key]), 'source': 'profiles/' + name}
        except Exception:
            pass
    return None


_BACKEND = 'nccl'


@contextlib.contextmanager
def _stdout_to_stderr():
    """File-descriptor level: whatever native libraries print to stdout inside the block goes to stderr, so that the
    one JSON line stays the only thing on rank 0's stdout."""
    import ctypes
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)      # C stdio of native libraries (RCCL's banner): drain it while fd 1 is still stderr
        except Exception:
            pass
        os.dup2(saved, 1)
        os.close(saved)


def _device(local_rank):
    assert torch.cuda.is_available(), 'bench.py needs a ROCm GPU'
    torch.cuda.set_device(local_rank)
    return torch.device('cuda', local_rank)


def _bind_thread(dev):
    torch.cuda.set_device(dev)


def _sync():
    torch.cuda.synchronize()


def _new_stream(dev):
    return torch.cuda.Stream(device=dev)


def _stream_ctx(stream):
    return torch.cuda.stream(stream)


def _gpu_elapsed_ms(fn):
    """GPU time of what `fn` enqueues on torch's current stream (the stream the engine launches on)."""
    st = torch.cuda.current_stream()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(st)
    fn()
    ev1.record(st)
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


# ---------------------------------------------------------------------------------------------------
# single-network measurements (latency, the other BASELINE configs)
# ---------------------------------------------------------------------------------------------------
def single_network_pass(proto, sweeps, reps=4, warm=2, pinned=False):
    """Wall time of ONE network's LE(sweeps)+BC pass with nothing else in flight, and its two halves as GPU time.
    `pinned`: run exactly `sweeps` sweeps (a network whose reference loop does not terminate)."""
    kw = dict(converge_thres=-1.0, converge_count=10 ** 9) if pinned else {}
    units = [make_unit([proto]) for _ in range(reps + warm + 2)]

    def one(u):
        u['le'].enqueue(sweeps, restart=True, max_sweeps=sweeps, **kw)
        u['bc'].run()
    for u in units[:warm]:
        one(u)
    _sync()
    t0 = time.perf_counter()
    for u in units[warm:warm + reps]:
        one(u)
    _sync()
    wall_ms = (time.perf_counter() - t0) * 1e3 / reps
    le_ms = _gpu_elapsed_ms(lambda: units[-2]['le'].enqueue(sweeps, restart=True, max_sweeps=sweeps, **kw))
    bc_ms = _gpu_elapsed_ms(lambda: units[-2]['bc'].run())
    for u in units:                      # a failed in-launch wait would have left the weights short of the result: raise
        u['le'].query()
        u['bc'].status()
    le, bc = units[0]['le'], units[0]['bc']
    streaming_bytes = pass_bytes(le, bc, sweeps)
    tiles = le.resident_tiles
    if tiles > 0:
        # the whole loop is ONE persistent launch: every paired weight is read once and written once, whatever the sweep count
        nbytes = 8 * le.rw_elements + 8 * bc.weight_elements
        engine = 'resident: one persistent launch of {} workgroups keeps the paired layers in LDS for all sweeps'.format(tiles)
    else:
        nbytes = streaming_bytes
        engine = 'streaming: one launch per sweep + convergence kernel ({})'.format(le.resident_reason)
    return dict(pass_ms=wall_ms, equalization_ms=le_ms, bias_correction_ms=bc_ms, algorithmic_bytes=nbytes,
                achieved_GBps=nbytes / (wall_ms * 1e-3) / 1e9, frac_of_hbm_peak=nbytes / (wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                streaming_formulation_bytes=streaming_bytes, engine=engine, resident_tiles=tiles)


def other_configs(spec, dev):
    """configs[2], configs[3] (single GPU): one network each, its own sweep count (or the pinned one)."""
    out = []
    for item in [s for s in spec.split(',') if s]:
        net, _, pin = item.partition(':')
        proto = prepare(net, 0, dev)
        n_w = sum(m.weight.numel() for m in proto[1].values() if type(m) in TARG)
        if pin:
            sweeps, pinned = int(pin), True
        else:
            probe = make_unit([proto])
            sweeps, pinned = probe['le'].run()['sweeps'], False
        m = single_network_pass(proto, sweeps, pinned=pinned)
        out.append({'net': net, 'weights': n_w, 'relations': len(proto[3]), 'sweeps': sweeps,
                    'sweeps_pinned': pinned, 'ms': m['pass_ms'], 'weights_per_s': n_w / (m['pass_ms'] * 1e-3),
                    'equalization_ms': m['equalization_ms'], 'bias_correction_ms': m['bias_correction_ms'],
                    'algorithmic_bytes': m['algorithmic_bytes'], 'achieved_GBps': m['achieved_GBps'],
                    'roofline_frac': m['frac_of_hbm_peak'], 'engine': m['engine'],
                    'streaming_formulation_GBps': m['streaming_formulation_bytes'] / (m['pass_ms'] * 1e-3) / 1e9})
    return out


def activation_range_kernels(shape, dev):
    """configs[4] (--distill_range): the kernels behind QuantMeasure.forward at MobileNetV2's largest activation
    (utils/quantize.py:102-119): per-sample min/max + running update (4 B per element), fake-quant with the recorded range
    (8 B per element) -> 12 B per element for the module; plus the weight quantiser of
    quantize_targ_layer over a whole MobileNetV2 (12 B per weight, two launches)."""
    from dfq_amd.utils import quantize as q
    g = torch.Generator().manual_seed(1)
    x = torch.randn(*shape, generator=g).clamp_(-2.1179, 2.64).to(dev)
    n = x.numel()
    out = torch.empty_like(x)
    running = torch.zeros(2, device=dev)
    rows = []

    def timed(fn, reps=10):
        """GPU time per call with `reps` calls enqueued back to back (the host prepares call i+1 while call i runs)."""
        fn()
        _sync()

        def many():
            for _ in range(reps):
                fn()
        return min(_gpu_elapsed_ms(many) for _ in range(3)) / reps
    ms = timed(lambda: q.sample_minmax_mean(x, shape[0], running=running))
    rows.append({'kernel': 'sample_minmax_kernel (+ sample_mean_kernel)', 'bytes': 4 * n, 'us': ms * 1e3})
    ms = timed(lambda: q.fake_quant_device(x, out, 8, False, 1, 0.0, 0.0, running, None))
    rows.append({'kernel': 'fake_quant_kernel', 'bytes': 8 * n, 'us': ms * 1e3})
    m = q.QuantMeasure(update_stat=True).to(dev).eval()
    ms = timed(lambda: m(x))
    rows.append({'kernel': 'QuantMeasure.forward (update_stat): 2 launches (per-sample extrema; mean + running range + quantise)', 'bytes': 12 * n, 'us': ms * 1e3})
    proto = prepare('mobilenet_v2' if n > 10 ** 6 else 'tiny_mobile', 0, dev)
    n_w = sum(m_.weight.numel() + (m_.bias.numel() if m_.bias is not None else 0) for m_ in proto[1].values() if type(m_) in TARG)
    import ctypes
    from dfq_amd import _ffi
    tensors = [t for m_ in proto[1].values() if type(m_) in TARG for t in ((m_.weight, 8), (m_.bias, 16)) if t[0] is not None]
    segs = (_ffi.DfqSegment * len(tensors))(*[_ffi.DfqSegment(t.data_ptr(), t.numel(), bits, 0, None) for t, bits in tensors])
    plan = ctypes.c_void_p()
    _ffi.check(_ffi.lib().dfq_quant_plan_create(segs, len(tensors), ctypes.byref(plan)))
    ms = timed(lambda: _ffi.check(_ffi.lib().dfq_quant_plan_run(plan, _ffi.stream_arg())))
    _ffi.lib().dfq_quant_plan_destroy(plan)
    rows.append({'kernel': 'seg_minmax_kernel + seg_fake_quant_kernel (the two launches of quantize_targ_layer over one network: '
                           '{:.1f} MB, cache-resident)'.format(4 * n_w / 1e6), 'bytes': 12 * n_w, 'us': ms * 1e3})
    for r in rows:
        r['GBps'] = r['bytes'] / max(r['us'], 1e-9) / 1e3
        r['frac'] = r['GBps'] / HBM_PEAK_GBS
    return {'config': 'MobileNetV2 --distill_range (configs[4]): activation [{}] float32'.format(', '.join(map(str, shape))),
            'elements': n, 'kernels': rows}


# ---------------------------------------------------------------------------------------------------
# opt-in: the same batch through the lazy-scale formulation (its own byte count, its own roofline entry)
# ---------------------------------------------------------------------------------------------------
def lazy_scale_pass(protos, net_sweeps, steps, warm, net='mobilenet_v2'):
    """SURVEY 7.3 item 9 / 8d "alternative byte count": LE of the batch with every network's sweep count GIVEN (what the
    reference's loop needs for it), computed from the pristine weights and the cumulative scales -- a sweep reads 4 B per
    paired element, the tensors are written once (8 B per weight) -- followed by the same bias correction.  Within 1e-5 of the
    default engine's tensors (tests/test_full_reference.py), not bit-identical: reported next to `value`, never as `value`."""
    from dfq_amd import dfq
    units = []
    for _ in range(steps + warm + 1):
        nets = [copy.deepcopy(p) for p in protos]
        units.append(dict(nets=nets, le=dfq.LazyLEPlan([(g, r) for (_, g, _, r) in nets], TARG),
                          bc=dfq.build_bc_plan_batch([(g, b) for (_, g, b, _) in nets], TARG)))

    def step(u):
        u['le'].run(net_sweeps)
        u['bc'].run()
    for u in units[:warm]:
        step(u)
    _sync()
    t0 = time.perf_counter()
    for u in units[warm:warm + steps]:
        step(u)
    _sync()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    le_ms = _gpu_elapsed_ms(lambda: units[-1]['le'].run(net_sweeps))
    for u in units:
        u['bc'].status()
    le = units[0]['le']
    n_w = sum(m.weight.numel() for m in protos[0][1].values() if type(m) in TARG)
    paired_per_net = le.paired_elements // len(protos)
    every_per_net = le.sweep_elements // len(protos)
    # every network reads all its paired layers in its first sweep and the interior, non-depthwise ones in every later sweep
    sweep_bytes = 4 * sum(paired_per_net + every_per_net * max(0, n - 1) for n in net_sweeps)
    final_bytes = 8 * le.weight_elements
    launches = 2 * le.levels * max(net_sweeps) + 1
    gbps = (sweep_bytes + final_bytes) / (le_ms * 1e-3) / 1e9
    return {'ms_per_step': ms, 'value': n_w * len(protos) / (ms * 1e-3), 'unit': 'weights/s', 'equalization_ms': le_ms,
            'sweeps': net_sweeps, 'sweeps_given': True, 'levels': le.levels, 'launches_per_pass': launches,
            'roofline': {'bound': 'hbm', 'kernel': 'lz_stats_kernel (+ lz_solve_kernel, rebuild_kernel): the whole lazy pass',
                         'achieved': gbps, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbps / HBM_PEAK_GBS,
                         'bytes_per_pass': sweep_bytes + final_bytes, 'bytes_per_network_first_sweep': 4 * paired_per_net,
                         'bytes_per_network_later_sweep': 4 * every_per_net,
                         'us_per_sweep': (le_ms * 1e3) / max(1, max(net_sweeps)),
                         'traffic': _pmc_kernel_traffic('lz_stats_kernel', net, len(protos))},
            'what': 'opt-in lazy-scale equalisation (csrc/dfq_le_lazy.hip): {} networks, sweep counts given per network (those of the '
                    'reference loop), read-only sweeps over W0 with the cumulative scales applied on the fly (4 B per paired element in the first sweep, then only the '
                    'non-depthwise layers in the interior of a chain: the extrema of the other passes are sweep-invariant and rescaled), '
                    'one final materialisation (8 B per weight), then the same bias correction; tensors within 1e-5 of the default '
                    'engine, not bit-identical'.format(len(protos))}


# ---------------------------------------------------------------------------------------------------
# config 5 end to end: update_quant_range over the distilled batches through the whole quantised network
# ---------------------------------------------------------------------------------------------------
def distill_range_pass(net, shape, n_batches, dev, group=None, world=1):
    """configs[4] as improve_dfq.py:280-297 runs it: `n_batches` batches of `shape` through the whole quantised network with
    every QuantMeasure recording its range (set_update_stat -> update_quant_range), the data resident on the GPU.  The
    convolutions are MIOpen's (the inference path, out of scope); the QuantMeasure kernels' share is the difference to the same
    forward passes with the quantisers switched to pass-through, priced at 12 B per element they see."""
    from dfq_amd import fxgraph, improve_dfq, synthetic
    from dfq_amd.utils import layer_transform as lt
    from dfq_amd.utils import quantize as q
    model, graph, bottoms = synthetic.build(net, seed=0)
    lt.merge_batchnorm(model, graph, bottoms, TARG)
    swapped = improve_dfq._swap_modules(model, {nn.Conv2d: q.QuantNConv2d, nn.Linear: q.QuantNLinear})
    for k in graph:
        if not isinstance(graph[k], str) and graph[k] in swapped:
            graph[k] = swapped[graph[k]]
    qmodel, graph, bottoms, tq = fxgraph.quantize_tensor_ops(model)      # + the quantisers of add / mean (layer_transform.py:10-14)
    qmodel.to(dev).eval()
    measures = [m for m in qmodel.modules() if isinstance(m, q.QuantMeasure)]
    g = torch.Generator().manual_seed(1)
    data = [torch.randn(*shape, generator=g).clamp_(-2.1179, 2.64).to(dev) for _ in range(n_batches)]
    seen = [0]
    hooks = [m.register_forward_pre_hook(lambda mod, args: seen.__setitem__(0, seen[0] + args[0].numel())) for m in measures]
    improve_dfq.set_update_stat(qmodel, [q.QuantMeasure], True)
    with torch.no_grad():
        qmodel(data[0])                                                  # warm-up: MIOpen picks its kernels
    elements_per_batch = seen[0]
    for h in hooks:
        h.remove()
    _sync()

    def run_all():
        # world > 1: the batches are split over the ranks and ONE all_reduce merges the [modules, 2] range table
        improve_dfq.update_quant_range(qmodel, data, graph, bottoms, group=group if world > 1 else None)
    with_q = min(_gpu_elapsed_ms(run_all) for _ in range(2))
    t0 = time.perf_counter()
    run_all()
    _sync()
    wall = (time.perf_counter() - t0) * 1e3
    # the same forwards with every quantiser a pass-through: what the convolutions alone cost
    saved = [m.forward for m in measures]
    for m in measures:
        m.forward = lambda x: x
    without_q = min(_gpu_elapsed_ms(run_all) for _ in range(2))
    for m, f in zip(measures, saved):
        m.forward = f
    improve_dfq.set_update_stat(qmodel, [q.QuantMeasure], False)
    qm_ms = max(with_q - without_q, 1e-9)
    nbytes = 12 * elements_per_batch * n_batches
    return {'net': net, 'batches': n_batches, 'batch_shape': list(shape), 'quant_measures': len(measures), 'world': world,
            'batches_per_rank': -(-n_batches // world), 'collectives_per_pass': 1 if world > 1 else 0,
            'elements_per_batch': elements_per_batch, 'ms_per_batch': with_q / n_batches, 'ms_total': with_q, 'wall_ms_total': wall,
            'convolutions_only_ms_per_batch': without_q / n_batches,
            'quant_measure_ms_per_batch': qm_ms / n_batches, 'quant_measure_share': qm_ms / with_q,
            'quant_measure_bytes_per_batch': 12 * elements_per_batch, 'quant_measure_GBps': nbytes / (qm_ms * 1e-3) / 1e9,
            'quant_measure_frac_of_hbm_peak': nbytes / (qm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'what': 'improve_dfq.update_quant_range over {} batches of {} through the whole quantised {} ({} QuantMeasure modules '
                    'incl. the tensor-op quantisers; improve_dfq.py:280-297); convolutions are MIOpen\'s, the QuantMeasure share is '
                    'the difference to the same forwards with pass-through quantisers'.format(n_batches, list(shape), net, len(measures))}


# ---------------------------------------------------------------------------------------------------
# the reference's default flow: a CPU-resident model through the drop-in entry points (PCIe inclusive)
# ---------------------------------------------------------------------------------------------------
def pcie_inclusive_pass(net, reps=3):
    """main_cls.py:149-181 with the model where the reference keeps it -- on the CPU: every entry point shadows the tensors it
    touches with device copies and writes its results back (dfq_amd._ffi.Stage).  Since round 6 the shadows outlive the call
    (the thread's persistent stage: consecutive plain calls transfer the network once, find their plans in the cache and bring
    back only what they rewrote).  Wall time of cross_layer_equalization + bias_correction, transfers and host work included.
    Never `value`."""
    from dfq_amd import dfq, synthetic
    from dfq_amd.utils import layer_transform as lt
    from dfq_amd.utils import relation as rel
    import dfq_amd
    best, scoped = None, None
    for _ in range(reps):
        model, graph, bottoms = synthetic.build(net, seed=0)             # stays on the CPU
        with _stdout_to_stderr():
            t0 = time.perf_counter()
            lt.merge_batchnorm(model, graph, bottoms, TARG)
            t1 = time.perf_counter()
            rels = rel.create_relation(graph, bottoms, TARG, delete_single=False)
            t2 = time.perf_counter()
            dfq.cross_layer_equalization(graph, rels, TARG)
            _sync()
            t3 = time.perf_counter()
            dfq.bias_correction(graph, bottoms, TARG)
            _sync()
            t4 = time.perf_counter()
        rec = {'merge_batchnorm_ms': (t1 - t0) * 1e3, 'create_relation_ms': (t2 - t1) * 1e3, 'equalization_ms': (t3 - t2) * 1e3,
               'bias_correction_ms': (t4 - t3) * 1e3, 'le_plus_bc_ms': (t4 - t2) * 1e3}
        if best is None or rec['le_plus_bc_ms'] < best['le_plus_bc_ms']:
            best = rec
    # the same two calls inside `with dfq_amd.staging():` -- one transfer each way for the sequence, plans keyed on the scope's
    # device copies (a service calibrating model after model of one architecture finds them in the plan cache)
    for _ in range(reps + 1):
        model, graph, bottoms = synthetic.build(net, seed=0)
        with _stdout_to_stderr():
            lt.merge_batchnorm(model, graph, bottoms, TARG)
            rels = rel.create_relation(graph, bottoms, TARG, delete_single=False)
            t2 = time.perf_counter()
            with dfq_amd.staging():
                dfq.cross_layer_equalization(graph, rels, TARG)
                dfq.bias_correction(graph, bottoms, TARG)
            _sync()
            t4 = time.perf_counter()
        scoped = (t4 - t2) * 1e3 if scoped is None else min(scoped, (t4 - t2) * 1e3)
    n_w = sum(m.weight.numel() for m in graph.values() if type(m) in TARG)
    best.update({'net': net, 'weights': n_w, 'weights_per_s': n_w / (best['le_plus_bc_ms'] * 1e-3),
                 'le_plus_bc_in_one_staging_scope_ms': scoped,
                 'what': 'CPU-resident {}: cross_layer_equalization + bias_correction through PLAIN calls of the drop-in entry points '
                         '(no staging() scope), wall time incl. transfers, host work and synchronisation (best of {}); the device copies '
                         'merge_batchnorm made are found again by the two calls (persistent stage, keyed on every tensor\'s _version / '
                         'data_ptr), each call writes back what it rewrote; '
                         'le_plus_bc_in_one_staging_scope_ms: the same two calls inside `with dfq_amd.staging():` (one transfer each '
                         'way, plans from the cache after the first model)'.format(net, reps)})
    return best


# ---------------------------------------------------------------------------------------------------
# config 4: one network sharded over the ranks (north_star's split)
# ---------------------------------------------------------------------------------------------------
def sharded_single_network(spec, steps, dev, dist, rank, world):
    """One entry of `sharded`: `net[:pinned sweeps]` -- the pinned pass (no exchange before the final all_gather) timed over
    `steps` passes, then the SAME network through the data-dependent mode (the reference's own loop, dfq.py:83-115: one 8-byte
    all_reduce of sum mean|dW| per sweep, the host in the loop) over min(steps, 3) passes."""
    from dfq_amd import dfq, sharded
    net, _, pin = spec.partition(':')
    sweeps = int(pin or 60)
    proto = prepare(net, 0, dev)                       # the SAME network on every rank (seed 0)
    n_w = sum(m.weight.numel() for m in proto[1].values() if type(m) in TARG)
    dd_steps = min(steps, 3)

    def fresh(n):
        reps = []
        for _ in range(n):
            model, graph, bottoms, rels = copy.deepcopy(proto)
            eq = sharded.ShardedEqualizer(graph, rels, TARG)            # partition + the owned components' plan (untimed)
            bc, _ = dfq.build_bc_plan(graph, bottoms, TARG)
            reps.append((eq, bc, graph))
        return reps

    def fence():
        _sync()
        dist.barrier()
        _sync()

    def timed(reps, warm, one):
        for r in reps[:warm]:
            one(r)
        fence()
        t0 = time.perf_counter()
        for r in reps[warm:]:
            one(r)
        fence()
        elapsed = time.perf_counter() - t0
        for r in reps:
            r[0].check()                         # an abandoned in-launch wait of any pass raises here
            r[1].status()
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) * 1e3 / (len(reps) - warm)

    def pinned(r):
        r[0].run(max_sweeps=sweeps, check=False)   # snapshot -> local sweeps -> ONE all_gather -> rebuild of every paired tensor
        r[1].run()                           # the correction chain is sequential over layers: replicated on every rank

    dd_sweeps = []

    def data_dependent(r):
        dd_sweeps.append(r[0].run(max_sweeps=None, check=False))       # chunks of sweeps, one all_reduce + one host read per chunk (sharded.py)
        r[1].run()

    reps = fresh(steps + 2)
    ms = timed(reps, 2, pinned)
    eq0 = reps[0][0]
    owner = list(eq0.owner)
    comps = sharded.relation_components(proto[3])
    per_rank = [0] * world
    for i, rr in enumerate(proto[3]):
        a, b, _ = rr.get_idxs()
        per_rank[owner[i]] += proto[1][a].weight.numel() + proto[1][b].weight.numel()
    exchange = eq0.exchange_bytes
    for r in reps:
        r[0].close()
    del reps
    reps = fresh(dd_steps + 1)
    dd_ms = timed(reps, 1, data_dependent)
    for r in reps:
        r[0].close()
    del reps
    return {'net': net, 'weights': n_w, 'relations': len(proto[3]), 'components': len(comps), 'sweeps': sweeps, 'sweeps_pinned': True,
            'world': world, 'ranks_owning_components': len(set(owner)), 'paired_elements_per_rank': per_rank,
            'ms_per_pass': ms, 'value': n_w / (ms * 1e-3), 'unit': 'weights/s',
            'scaling': 'strong', 'collectives_per_pass': 1, 'exchange_bytes_per_rank': exchange,
            'data_dependent_ms': dd_ms, 'data_dependent_sweeps': dd_sweeps[-1], 'data_dependent_chunk': sharded.ShardedEqualizer.CHUNK,
            'data_dependent_collectives_per_pass': -(-dd_sweeps[-1] // sharded.ShardedEqualizer.CHUNK) + 1,
            'backend': dist.get_backend(),
            'what': 'ONE {} network per pass: relation components partitioned over the ranks (greedy by paired elements: '
                    'paired_elements_per_rank), {} pinned sweeps per rank on the '
                    'owned components (on scratch copies), one all_gather of the cumulative scale vectors, ONE batched rebuild launch of every '
                    'paired tensor on every rank (W = diag(S_out) W0 diag(1/S_in): all ranks end bit-identical), bias correction replicated on '
                    'every rank; plans prebuilt, weights resident.  data_dependent_ms: the same pass with the reference\'s own stopping '
                    'rule (dfq.py:83-115) -- chunks of data_dependent_chunk sweeps on the device, ONE all_reduce of a chunk\'s per-sweep '
                    'sums of mean|dW|, the verdicts drawn on the device, one host read per chunk; a loop that stops inside a chunk goes back to '
                    'the chunk\'s start (scratch copies) and runs exactly the sweeps that happen'.format(net, sweeps)}


# ---------------------------------------------------------------------------------------------------
def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node {}'.format(args.gpus))
    dev = _device(local_rank)
    dist = None
    if world > 1 or args.sharded:
        import torch.distributed as dist_mod
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if world == 1:                                     # the sharded pass needs a (one-rank) group
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', str(_free_port()))
        try:
            with _stdout_to_stderr():                          # RCCL prints a version banner to stdout at start-up
                if _BACKEND == 'nccl':                         # RCCL over xGMI: one rank per GPU
                    dist_mod.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
                else:                                          # CPU dry run of the multi-rank path (tests/emu/dryrun.py)
                    dist_mod.init_process_group(_BACKEND, rank=rank, world_size=world)
                dist_mod.barrier()                             # communicators are created lazily: do it here
            dist = dist_mod
        except Exception as e:                                 # N = 1 only: the headline does not need a group
            if world > 1:
                raise
            print('bench.py: no process group at N=1 ({}); the sharded entry is skipped'.format(e), file=sys.stderr)

    from dfq_amd import _ffi
    _ffi.lib()

    # `--batch B` distinct networks (different seeds) are calibrated together in every step
    batch = max(1, args.batch)
    protos = [prepare(args.net, seed=rank * 1000 + i, dev=dev) for i in range(batch)]
    n_w = sum(m.weight.numel() for m in protos[0][1].values() if type(m) in TARG)
    n_layers = sum(1 for m in protos[0][1].values() if type(m) in TARG)

    # sweep counts of the reference's convergence loop on these inputs (device-side loop, untimed); the
    # timed steps enqueue the largest one, every network still stops at its own count
    probe = make_unit(protos)
    probe['le'].run()
    net_sweeps = [r['sweeps'] for r in probe['le'].query_all()[0]]
    sweeps = args.sweeps if args.sweeps > 0 else max(net_sweeps)
    if world > 1 and args.sweeps == 0:          # every rank enqueues the same amount of work per step
        t = torch.tensor([sweeps], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sweeps = int(t.item())
    levels = probe['le'].levels
    paired = probe['le'].paired_elements                    # sum over relations of n1 + n2
    rw, ro = probe['le'].rw_elements, probe['le'].ro_elements
    sweep_bytes, defer_depth, deferred = probe['le'].sweep_bytes, probe['le'].defer_depth, probe['le'].deferred_elements
    fr_elems, fr_group = probe['le'].free_running_elements, probe['le'].free_running_group     # (dfq_le_cf.hpp)

    units = [make_unit(protos) for _ in range(args.steps + args.warmup)]

    force = dict(converge_thres=-1.0, converge_count=10 ** 9) if args.force_sweeps else {}

    def step(u):
        u['le'].enqueue(sweeps, restart=True, max_sweeps=sweeps, **force)
        u['bc'].run()

    def fence():
        _sync()
        if world > 1:
            dist.barrier()
        _sync()

    # Units are independent jobs: `--streams S` keeps S of them in flight, each on its own HIP stream fed
    # by its own host thread (ctypes releases the GIL).
    n_streams = max(1, args.streams)
    streams = [_new_stream(dev) for _ in range(n_streams)]

    def run(work):
        if n_streams == 1:
            with _stream_ctx(streams[0]):
                for u in work:
                    step(u)
            return
        import threading

        def worker(i):
            _bind_thread(dev)                  # HIP's current device is per host thread; new threads start on device 0
            with _stream_ctx(streams[i]):
                for u in work[i::n_streams]:
                    step(u)
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(n_streams)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()

    run(units[:args.warmup])
    fence()
    t0 = time.perf_counter()
    run(units[args.warmup:])
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not args.force_sweeps and args.sweeps == 0:
        # every network of EVERY timed batch stopped where the probe stopped (query also surfaces a failed in-launch wait)
        for u in units[args.warmup:]:
            done = [r['sweeps'] for r in u['le'].query_all()[0]]
            assert done == net_sweeps, 'timed steps ran {} sweeps, the probe {}'.format(done, net_sweeps)
    for u in units[args.warmup:]:
        u['bc'].status()
    ms_per_step = elapsed * 1e3 / args.steps

    # where a step's time goes: the two halves of one unit on an otherwise idle GPU (one stream)
    br_units = [make_unit(protos) for _ in range(3)]
    with _stream_ctx(streams[0]):
        for u in br_units[:1]:
            step(u)
        _sync()
        le_ms = sum(_gpu_elapsed_ms(lambda u=u: u['le'].enqueue(sweeps, restart=True, max_sweeps=sweeps, **force))
                    for u in br_units[1:]) / 2
        bc_ms = sum(_gpu_elapsed_ms(lambda u=u: u['bc'].run()) for u in br_units[1:]) / 2

    out = {
        'metric': 'conv weights calibrated/sec (LE+BC pass, MobileNetV2)' if args.net == 'mobilenet_v2'
                  else 'conv weights calibrated/sec (LE+BC pass, {})'.format(args.net),
        'value': n_w * batch * world / (ms_per_step * 1e-3),
        'unit': 'weights/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': ms_per_step,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {
            'workload': '{} --relu --equalize --correction: {} layers, {} weights, {} relations per network; a step '
                        'calibrates a batch of {} networks (distinct random seeds) per GPU: LE until each network\'s '
                        'reference convergence test fires (thres 2e-7; {} sweeps) + bias correction; weights resident '
                        'in HBM'.format(args.net, n_layers, n_w, len(protos[0][3]), batch, sorted(set(net_sweeps))),
            'le_sweeps': net_sweeps,
            'networks_per_step': batch * world,
            'launches_per_sweep': levels + 1,
            'units_in_flight_per_gpu': n_streams,
            'one_unit_alone_ms': {'equalization': le_ms, 'bias_correction': bc_ms},
            # host-side, once per batch, outside the timed region (like graph tracing / BN folding): descriptor and
            # launch tables of the two plans, small device allocations, one synchronisation
            'plan_build_ms_per_unit': sum(u['plan_build_ms'] for u in units) / len(units),
            'batch_layout': ('one allocation, fixed stride per network (dfq_amd/arena.py NetworkBatch); a plan = the first network\'s '
                             'tables + {} base addresses'.format(batch)) if ARENA and batch > 1 else 'tensors where torch allocated them',
            'batch_layout_ms_per_unit': sum(u.get('layout_ms', 0.0) for u in units) / len(units),
        },
    }

    # host-inclusive rate: a calibration service that builds a plan per batch pays the table building on the host (Python: one
    # structure template per architecture, device addresses gathered per network) before it can enqueue -- no overlap assumed
    pb = out['config']['plan_build_ms_per_unit']
    out['host_inclusive'] = {'plan_build_ms_per_unit': pb, 'ms_per_step_with_plan_build': ms_per_step + pb,
                             'value': n_w * batch * world / ((ms_per_step + pb) * 1e-3), 'unit': 'weights/s',
                             'what': 'the two plans of a batch created on the host for every batch (one network\'s tables + a base address per '
                                     'network; device tables from a free list), then the timed step; excludes fx tracing, BN folding, relation '
                                     'pairing and laying the batch out as one allocation (once per architecture / per network / per batch, '
                                     'when the models are loaded: batch_layout_ms_per_unit)'}
    if ARENA and batch > 1 and rank == 0:
        # the same plans over tensors left where torch allocated them: 7 000 addresses gathered in Python per batch
        from dfq_amd import dfq
        nets = [copy.deepcopy(p) for p in protos]
        _sync()
        t0 = time.perf_counter()
        le_g = dfq.build_le_plan_batch([(g, r) for (_, g, _, r) in nets], TARG)
        bc_g = dfq.build_bc_plan_batch([(g, b) for (_, g, b, _) in nets], TARG)
        out['host_inclusive']['plan_build_ms_per_unit_scattered_tensors'] = (time.perf_counter() - t0) * 1e3
        le_g.close()
        bc_g.close()

    # ---- one network alone: the latency a caller of the drop-in API sees (first-class, with its own roofline fraction) ----
    if rank == 0:
        with _stream_ctx(streams[0]):
            lat = single_network_pass(protos[0], net_sweeps[0] if args.sweeps == 0 else sweeps, pinned=args.sweeps > 0)
        out['latency'] = {
            'single_network_pass_ms': lat['pass_ms'], 'sweeps': net_sweeps[0] if args.sweeps == 0 else sweeps,
            'equalization_gpu_ms': lat['equalization_ms'], 'bias_correction_gpu_ms': lat['bias_correction_ms'],
            'weights_per_s': n_w / (lat['pass_ms'] * 1e-3), 'algorithmic_bytes_per_pass': lat['algorithmic_bytes'],
            'achieved_GBps': lat['achieved_GBps'], 'frac_of_hbm_peak': lat['frac_of_hbm_peak'],
            'engine': lat['engine'],
            # what the same pass would move sweep by sweep (the streaming kernel's accounting), for comparison with round 1
            'streaming_formulation_bytes_per_pass': lat['streaming_formulation_bytes'],
            'streaming_formulation_GBps': lat['streaming_formulation_bytes'] / (lat['pass_ms'] * 1e-3) / 1e9,
            'bound': 'latency: a chain of per-sweep statistics hand-offs between workgroups (a network is {:.1f} MB; with the '
                     'weights resident on-chip HBM sees them once, so the HBM fraction says how far from bandwidth-bound a single '
                     'network is, not how good the kernel is; DESIGN.md 4.2)'.format(n_w * 4 / 1e6),
        }
        out['config']['single_pass_latency_ms'] = lat['pass_ms']
        # like for like with BASELINE.json's metric (ONE network per pass, configs[1]); `value` above is a batch of
        # independent networks per step -- an API the reference does not have
        out['value_single_network'] = n_w / (lat['pass_ms'] * 1e-3)
        out['value_single_network_what'] = ('weights/s of ONE {} through LE + BC with nothing else in flight ({:.3f} ms per pass); '
                                            'latency-bound, see `latency`'.format(args.net, lat['pass_ms']))

    if rank == 0 and not args.no_roofline:
        # Dominant kernel: le_level_kernel (one launch per sweep).  Algorithmic bytes of a launch = 8 B per
        # element it reads and writes (every weight of a paired layer once per sweep; the ranges are
        # by-products) + 4 B per element of an interior layer it only measures.  Its duration
        # comes from HIP events on the launch stream:
        #   (a) one event pair around a whole run of `sweeps` sweeps  -> wall time per sweep;
        #   (b) event pairs around every single launch (dfq_le_profile), minus the same pair around
        #       nothing, -> how that wall time splits between the level launches and the convergence
        #       kernel.  (a) x share(b) is what rocprofv3 reports as the kernel's average duration.
        prof_rep = make_unit(protos)
        prof = prof_rep['le'].profile(sweeps, max_sweeps=sweeps, **force)
        empty = prof['empty_bracket_ms']
        lvl_corr = [max(ms / sweeps - empty, 0.0) for ms in prof['level_ms']]          # per launch, ms
        ctl_corr = max(prof['control_ms'] / sweeps - empty, 0.0)
        # the lean launch of the free-running layers runs once per group of sweeps: its time per launch and per sweep
        n_lean = prof.get('lean_launches', 0)
        lean_launch_corr = max(prof['lean_ms'] / n_lean - empty, 0.0) if n_lean else 0.0
        lean_corr = lean_launch_corr * n_lean / sweeps
        total_corr = max(sum(lvl_corr) + ctl_corr + lean_corr, 1e-12)
        share_levels = sum(lvl_corr) / total_corr
        # every sweep must do real work here, so the convergence exit is disabled for this run
        always = dict(converge_thres=-1.0, converge_count=10 ** 9)
        wall_rep = make_unit(protos)
        wall_rep['le'].enqueue(0, restart=True, max_sweeps=sweeps, **always)           # restart outside the bracket
        _sync()
        sweep_ms = _gpu_elapsed_ms(lambda: wall_rep['le'].enqueue(sweeps, restart=False, max_sweeps=sweeps, **always)) / sweeps
        assert wall_rep['le'].query()['sweeps'] == sweeps
        launches = sweeps * levels
        avg_ms = sweep_ms * share_levels / levels
        # bytes as executed: one-way-scaled layers are read every sweep and stored every `defer_depth`-th one (dfq_le.hip,
        # "Deferred stores"); the timed bracket includes the launch that brings them up to date at the end
        # ... and the free-running layers (8 B per element once per group of sweeps) are le_lean_kernel's bytes, not this kernel's
        bytes_per_sweep = sweep_bytes - 8.0 * fr_elems / fr_group
        avg_bytes = bytes_per_sweep / levels
        achieved = avg_bytes / max(avg_ms * 1e-3, 1e-12) / 1e9
        per_level = []
        for l in range(levels):
            info = prof_rep['le'].level_info(l)
            us = sweep_ms * 1e3 * lvl_corr[l] / total_corr
            nbytes = bytes_per_sweep if levels == 1 else 8 * info['rw_elements'] + 4 * info['ro_elements']
            per_level.append({'level': l, 'relations': info['relations'], 'workgroups': info['workgroups'],
                              'bytes': nbytes, 'us': us, 'GBps': nbytes / max(us, 1e-9) / 1e3})
        # SURVEY.md 8(d)'s eager contract for the same launches: 8 B per paired element + 12 B per weight for a separate
        # convergence pass with a snapshot refresh -- what a restatement of dfq.py:84-108 would move.  This engine takes the
        # diff inside the rescale pass and writes interior layers once, so it moves 2.4x fewer bytes; `achieved` / `frac` are
        # priced on the bytes it actually needs.  The eager figure is NOT a roofline fraction of this kernel: it says how fast
        # an eager implementation would have to stream to finish a sweep in the same time.
        survey_bytes = (8 * paired + 12 * n_w * batch) / levels
        traffic, traffic_src = _pmc_traffic(args.net, batch)
        out['roofline'] = {
            'bound': 'hbm', 'kernel': 'le_level_kernel', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_unit': 'bytes per launch',
            'traffic_source': ('{}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this command at this batch size; '
                               'a profiler pass cannot run inside this process'.format(traffic_src)) if traffic_src else None,
            'bytes_per_launch': avg_bytes, 'us_per_launch': avg_ms * 1e3, 'launches_timed': launches,
            'deferred_store_depth': defer_depth, 'deferred_elements_per_launch': deferred / levels,
            'bytes_per_launch_storing_every_sweep': (8 * rw + 4 * ro) / levels,
            'equivalent_GBps_storing_every_sweep': (8 * rw + 4 * ro) / levels / max(avg_ms * 1e-3, 1e-12) / 1e9,
            'eager_formulation_bytes_per_launch': survey_bytes,
            'eager_formulation_equivalent_GBps': survey_bytes / max(avg_ms * 1e-3, 1e-12) / 1e9,
            'sweep_wall_us': sweep_ms * 1e3, 'control_us_per_sweep': sweep_ms * 1e3 * ctl_corr / total_corr,
            'event_pair_overhead_us': empty * 1e3, 'levels': per_level,
            'bytes_per_sweep_all_kernels': sweep_bytes,
            'GBps_per_sweep_all_kernels': sweep_bytes / max(sweep_ms * 1e-3, 1e-12) / 1e9,
        }
        if n_lean:
            # the second streaming kernel of a sweep: one launch per `fr_group` sweeps over the layers whose statistics are
            # closed-form (read once, written once, |dW| of the whole group formed on the way)
            lean_us = sweep_ms * 1e3 * (lean_corr / total_corr) * sweeps / n_lean
            out['roofline']['free_running'] = {
                'kernel': 'le_lean_kernel', 'sweeps_per_launch': fr_group, 'launches_timed': n_lean,
                'bytes_per_launch': 8.0 * fr_elems, 'us_per_launch': lean_us, 'us_per_sweep': lean_us / fr_group,
                'achieved': 8.0 * fr_elems / max(lean_us * 1e-6, 1e-12) / 1e9, 'unit': 'GB/s',
                'frac': 8.0 * fr_elems / max(lean_us * 1e-6, 1e-12) / 1e9 / HBM_PEAK_GBS,
            }

    # ---- the other BASELINE configurations, each with ms and roofline fraction.  These legs come after the headline has been
    #      measured: one of them failing is recorded in the line (`<name>_error`) instead of costing the whole line ----
    def side_leg(name, fn):
        try:
            with _stream_ctx(streams[0]):
                return fn()
        except Exception as e:                                  # noqa: BLE001 -- reported, not swallowed
            import traceback
            traceback.print_exc(file=sys.stderr)
            out[name + '_error'] = repr(e)
            return None
    if rank == 0 and args.others:
        res = side_leg('others', lambda: other_configs(args.others, dev))
        if res is not None:
            out['config']['others'] = res
    if rank == 0 and args.act_shape:
        res = side_leg('activation_ranges', lambda: activation_range_kernels([int(v) for v in args.act_shape.split(',')], dev))
        if res is not None:
            out['config']['activation_ranges'] = res

    if rank == 0 and args.lazy_steps > 0 and args.sweeps == 0:
        res = side_leg('lazy_scale', lambda: lazy_scale_pass(protos, net_sweeps, args.lazy_steps, 2, net=args.net))
        if res is not None:
            out['lazy_scale'] = res
            out['value_lazy_scale'] = res['value']
    if args.distill and (rank == 0 or world > 1):
        # every rank takes part at N > 1: data-parallel over the distilled batches, one all_reduce of the range table
        dnet, dn, dshape = args.distill.split(':')
        if world > 1:
            with _stream_ctx(streams[0]):
                res = distill_range_pass(dnet, [int(v) for v in dshape.split(',')], int(dn), dev, group=dist.group.WORLD, world=world)
        else:
            res = side_leg('distill_range', lambda: distill_range_pass(dnet, [int(v) for v in dshape.split(',')], int(dn), dev))
        if res is not None and rank == 0:
            out['config']['distill_range'] = res
    if rank == 0 and world == 1 and args.pcie:
        rec = side_leg('pcie_inclusive', lambda: pcie_inclusive_pass(args.pcie))
        if rec is not None:
            out['pcie_inclusive'] = rec
            out['pcie_inclusive_ms'] = rec['le_plus_bc_ms']

    # ---- config 4 as north_star splits it (every rank takes part) ----
    if args.sharded and dist is not None:
        with _stream_ctx(streams[0]):
            sh = [sharded_single_network(item, args.sharded_steps, dev, dist, rank, world) for item in args.sharded.split(',') if item]
        if rank == 0:
            out['sharded'] = sh

    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        out['cpu_baseline'], cpu_sweeps = cpu_baseline(args.net, rank * 1000, args.cpu_seconds)
        if args.sweeps == 0 and cpu_sweeps != net_sweeps[0]:      # a parity failure: say so IN the line (and fail after printing it)
            out['cpu_baseline_error'] = 'engine needed {} sweeps, the CPU oracle {}'.format(net_sweeps[0], cpu_sweeps)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and 'cpu_baseline_error' in out:
        raise SystemExit('bench.py: ' + out['cpu_baseline_error'])


if __name__ == '__main__':
    main()
