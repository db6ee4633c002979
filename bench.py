#!/usr/bin/env python
"""Headline benchmark: conv weights calibrated per second by one LE+BC pass over a synthetic
MobileNetV2 (BASELINE.json configs[1]: `--relu --equalize --correction`, 53 layers).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A step = one complete pass of the hot path over one network whose folded weights are already
resident in HBM: cross_layer_equalization for exactly the number of sweeps the reference's
convergence test needs on this input (measured once, untimed, by the device-side loop) followed by
bias_correction.  Every step runs on its own replica of the network (in-place algorithm, so a used
replica is "calibrated"; 288 GB of HBM hold hundreds of replicas), nothing is restored or skipped
inside the timed region and the host never synchronises between launches.

Multi-GPU: the unit of work is a network; rank r calibrates its own replicas (independent
objects, no data-path collective), `value` = weights calibrated by all ranks / max-over-ranks time
-> "scaling": "weak".  The sharded single-network mode with its RCCL exchange lives in
dfq_amd/sharded.py and is covered by tests, not by this line.

One JSON line on rank 0, with `roofline` (dominant kernel le_level_kernel: algorithmic bytes per
launch / HIP-event duration per launch) and `cpu_baseline` (the numpy oracle, i.e. a vectorised
CPU port of the reference's algorithm, timed on this host).
"""
from __future__ import annotations

import argparse
import contextlib
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch                      # noqa: E402
import torch.nn as nn             # noqa: E402

TARG = [nn.Conv2d, nn.Linear]
HBM_PEAK_GBS = 8000.0             # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--net', default='mobilenet_v2', choices=['mobilenet_v2', 'resnet18', 'deeplab_mnv2', 'tiny_mobile'])
    ap.add_argument('--sweeps', type=int, default=0, help='pin the LE sweep count (0 = what the convergence test needs)')
    ap.add_argument('--cpu-seconds', type=float, default=10.0, help='CPU-baseline budget (0 disables)')
    ap.add_argument('--batch', type=int, default=32, help='networks calibrated together in one step (one batched plan)')
    ap.add_argument('--streams', type=int, default=2, help='steps in flight per GPU (one HIP stream + host thread each)')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--force-sweeps', action='store_true', help='tuning: run --sweeps sweeps regardless of convergence')
    return ap.parse_args()


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


def make_unit(protos):
    """One unit of work = a batch of networks calibrated together: deep copies of the prototypes plus
    one LE plan and one BC plan over the whole batch (a batch of 1 is an ordinary single-network plan)."""
    from dfq_amd import dfq
    nets = [copy.deepcopy(p) for p in protos]
    t0 = time.perf_counter()
    if len(nets) == 1:
        model, graph, bottoms, rels = nets[0]
        le = dfq.build_le_plan(graph, rels, TARG)
        bc, _ = dfq.build_bc_plan(graph, bottoms, TARG)
    else:
        le = dfq.build_le_plan_batch([(g, r) for (_, g, _, r) in nets], TARG)
        bc = dfq.build_bc_plan_batch([(g, b) for (_, g, b, _) in nets], TARG)
    return dict(nets=nets, le=le, bc=bc, plan_build_ms=(time.perf_counter() - t0) * 1e3)


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
    return dict(value=n_w * reps / spent, unit='weights/s', cores=1, kind='port',
                sample='{} full LE({} sweeps)+BC passes of the numpy oracle over the same synthetic {} '
                       '({} weights), {:.1f} s of CPU time'.format(reps, sweeps, net, n_w, spent)), sweeps


def _pmc_traffic(net, batch):
    """HBM bytes per launch of le_level_kernel from the committed PMC summary (MobileNetV2 only), else null.
    The counters were collected on a batch of `summary['batch']` networks; a launch over `batch` networks runs
    the same tiles once per network, so the figure scales with the batch (returned with the batch it came from)."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_summary.json')
    if net != 'mobilenet_v2' or not os.path.exists(path):
        return None, None
    try:
        summary = json.load(open(path))
        return summary['le_level_kernel']['traffic_bytes_per_launch'] * batch / summary['batch'], summary['batch']
    except Exception:
        return None, None


_BACKEND = 'nccl'


@contextlib.contextmanager
def _stdout_to_stderr():
    """File-descriptor level: whatever native libraries print to stdout inside the block goes to stderr, so that the
    one JSON line stays the only thing on rank 0's stdout."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
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
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        with _stdout_to_stderr():                          # RCCL prints a version banner to stdout at start-up
            if _BACKEND == 'nccl':                         # RCCL over xGMI: one rank per GPU
                dist.init_process_group('nccl', device_id=dev)
            else:                                          # CPU dry run of the multi-rank path (tests/emu/dryrun.py)
                dist.init_process_group(_BACKEND)
            dist.barrier()                                 # communicators are created lazily: do it here

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
    if dist is not None and args.sweeps == 0:          # every rank enqueues the same amount of work per step
        t = torch.tensor([sweeps], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sweeps = int(t.item())
    levels = probe['le'].levels
    paired = probe['le'].paired_elements                    # sum over relations of n1 + n2
    rw, ro = probe['le'].rw_elements, probe['le'].ro_elements

    units = [make_unit(protos) for _ in range(args.steps + args.warmup)]

    force = dict(converge_thres=-1.0, converge_count=10 ** 9) if args.force_sweeps else {}

    def step(u):
        u['le'].enqueue(sweeps, restart=True, max_sweeps=sweeps, **force)
        u['bc'].run()

    def fence():
        _sync()
        if dist is not None:
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
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if not args.force_sweeps and args.sweeps == 0:
        done = [r['sweeps'] for r in units[-1]['le'].query_all()[0]]
        assert done == net_sweeps, 'timed steps ran {} sweeps, the probe {}'.format(done, net_sweeps)
    ms_per_step = elapsed * 1e3 / args.steps

    # latency of ONE single-network pass with nothing else in flight (reported next to the throughput)
    lat_units = [make_unit(protos[:1]) for _ in range(6)]
    with _stream_ctx(streams[0]):
        for u in lat_units[:2]:
            u['le'].enqueue(net_sweeps[0], restart=True, max_sweeps=net_sweeps[0])
            u['bc'].run()
        fence()
        t0 = time.perf_counter()
        for u in lat_units[2:]:
            u['le'].enqueue(net_sweeps[0], restart=True, max_sweeps=net_sweeps[0])
            u['bc'].run()
        fence()
        single_ms = (time.perf_counter() - t0) * 1e3 / 4

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
            'single_pass_latency_ms': single_ms,
            'one_unit_alone_ms': {'equalization': le_ms, 'bias_correction': bc_ms},
            # host-side, once per batch, outside the timed region (like graph tracing / BN folding): descriptor and
            # launch tables of the two plans, small device allocations, one synchronisation
            'plan_build_ms_per_unit': sum(u['plan_build_ms'] for u in units) / len(units),
        },
    }

    if rank == 0 and not args.no_roofline:
        # Dominant kernel: le_level_kernel (5 launches per sweep).  Algorithmic bytes of a launch = 8 B per
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
        share_levels = sum(lvl_corr) / max(sum(lvl_corr) + ctl_corr, 1e-12)
        # every sweep must do real work here, so the convergence exit is disabled for this run
        always = dict(converge_thres=-1.0, converge_count=10 ** 9)
        wall_rep = make_unit(protos)
        wall_rep['le'].enqueue(0, restart=True, max_sweeps=sweeps, **always)           # restart outside the bracket
        _sync()
        sweep_ms = _gpu_elapsed_ms(lambda: wall_rep['le'].enqueue(sweeps, restart=False, max_sweeps=sweeps, **always)) / sweeps
        assert wall_rep['le'].query()['sweeps'] == sweeps
        launches = sweeps * levels
        avg_ms = sweep_ms * share_levels / levels
        bytes_per_sweep = 8 * rw + 4 * ro
        avg_bytes = bytes_per_sweep / levels
        achieved = avg_bytes / max(avg_ms * 1e-3, 1e-12) / 1e9
        per_level = []
        for l in range(levels):
            info = prof_rep['le'].level_info(l)
            us = sweep_ms * 1e3 * lvl_corr[l] / max(sum(lvl_corr) + ctl_corr, 1e-12)
            nbytes = 8 * info['rw_elements'] + 4 * info['ro_elements']
            per_level.append({'level': l, 'relations': info['relations'], 'workgroups': info['workgroups'],
                              'bytes': nbytes, 'us': us, 'GBps': nbytes / max(us, 1e-9) / 1e3})
        # SURVEY.md 8(d) contract figure for the same launches: 8 B per paired element + 12 B per weight for the
        # convergence diff with a snapshot refresh -- what an eager restatement of dfq.py:84-108 moves.  This
        # engine takes the diff inside the rescale pass, so it moves less; `achieved` above is priced on the
        # bytes it actually needs, `achieved_survey_8d` on the contract figure.
        survey_bytes = (8 * paired + 12 * n_w * batch) / levels
        traffic, traffic_batch = _pmc_traffic(args.net, batch)
        out['roofline'] = {
            'bound': 'hbm', 'kernel': 'le_level_kernel', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_unit': 'bytes per launch',
            'traffic_source': 'profiles/r01_pmc_summary.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes on a '
                              'batch of {} networks, scaled to this batch (same tiles per network); a profiler pass '
                              'cannot run inside this process'.format(traffic_batch),
            'bytes_per_launch': avg_bytes, 'us_per_launch': avg_ms * 1e3, 'launches_timed': launches,
            'achieved_survey_8d': survey_bytes / max(avg_ms * 1e-3, 1e-12) / 1e9,
            'frac_survey_8d': survey_bytes / max(avg_ms * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS,
            'sweep_wall_us': sweep_ms * 1e3, 'control_us_per_sweep': sweep_ms * 1e3 * (1.0 - share_levels),
            'event_pair_overhead_us': empty * 1e3, 'levels': per_level,
        }
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        out['cpu_baseline'], cpu_sweeps = cpu_baseline(args.net, rank * 1000, args.cpu_seconds)
        if args.sweeps == 0:
            assert cpu_sweeps == net_sweeps[0], 'engine needed {} sweeps, the CPU oracle {}'.format(net_sweeps[0], cpu_sweeps)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
