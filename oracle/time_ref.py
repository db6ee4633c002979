"""Time the UNMODIFIED reference's CPU path (oracle/_ref, see oracle/build_ref.py) on THIS box's host cores.

TEST / MEASUREMENT INFRASTRUCTURE: the `cpu_baseline.reference` leg of bench.py runs this file in a subprocess
(so the reference's top-level `utils` package never enters the bench process) and reads the JSON it prints.

    python oracle/time_ref.py --net mobilenet_v2 --sweeps 2 [--threads N]

What is timed (SURVEY 8d "CPU baseline timing"): K sweeps of the reference's own `dfq._layer_equalization` driven over
the relation list exactly as dfq.py:85-101 does (the reference's loop has no sweep cap; its full data-dependent run is
47 sweeps = 75 s for MobileNetV2, too long for a default bench run -- `s_per_sweep` scales), plus the per-sweep deepcopy and
diff of dfq.py:84,105-108, then the reference's `dfq.bias_correction` (dfq.py:173-293) once.  BN folding and relation
pairing are the reference's own (`merge_batchnorm`, `create_relation`), untimed.  The synthetic network is the bench's.
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, 'oracle', '_ref')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--net', default='mobilenet_v2')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--sweeps', type=int, default=2)
    ap.add_argument('--full-sweeps', type=int, default=0, help='sweeps of a whole pass (for the extrapolated weights/s)')
    ap.add_argument('--threads', type=int, default=0, help='torch intra-op threads (default min(8, cores): the reference\'s path is a Python loop of '
                    'tiny per-channel ops; on the 256-thread GPU box torch.set_num_threads(256) was measured 4-5x SLOWER than 8)')
    args = ap.parse_args()
    if not os.path.isfile(os.path.join(REFDIR, 'dfq.pyc')):
        print(json.dumps({'error': 'oracle/_ref is not built (oracle/build_ref.py needs /root/reference)'}))
        return 2
    sys.path.insert(0, ROOT)
    sys.path.insert(0, REFDIR)
    sys.dont_write_bytecode = True
    import torch
    import torch.nn as nn
    import dfq as ref_dfq                                   # oracle/_ref/dfq.pyc: the reference, byte-compiled
    from utils import layer_transform as ref_lt
    from utils import relation as ref_rel
    assert os.path.dirname(os.path.abspath(ref_dfq.__file__)) == REFDIR, ref_dfq.__file__
    from dfq_amd import synthetic

    targ = [nn.Conv2d, nn.Linear]
    cores = args.threads or min(8, os.cpu_count())
    torch.set_num_threads(cores)
    _stdout = sys.stdout
    sys.stdout = open(os.devnull, 'w')                      # the reference prints progress lines
    try:
        model, graph, bottoms = synthetic.build(args.net, seed=args.seed)
        ref_lt.merge_batchnorm(model, graph, bottoms, targ)
        rels = ref_rel.create_relation(graph, bottoms, targ, delete_single=False)
        n_w = sum(m.weight.numel() for m in graph.values() if type(m) in targ)
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(args.sweeps):
                prev = copy.deepcopy(graph)                 # dfq.py:84
                for rr in rels:                             # dfq.py:85-101
                    lf, ls, bn = rr.get_idxs()
                    if graph[lf].bias is None:
                        graph[lf].bias = nn.Parameter(torch.zeros(graph[lf].weight.size(0)), requires_grad=False)
                    graph[lf].weight, graph[ls].weight, graph[lf].bias, S = ref_dfq._layer_equalization(
                        graph[lf].weight, graph[ls].weight, graph[lf].bias, graph[bn].fake_weight, graph[bn].fake_bias)
                    rr.set_scale_vec(S)
                diff = 0.0
                for k in graph:                             # dfq.py:105-108
                    if type(graph[k]) in targ:
                        diff += float(torch.mean(torch.abs(graph[k].weight - prev[k].weight)))
        t_le = time.perf_counter() - t0
        t0 = time.perf_counter()
        ref_dfq.bias_correction(graph, bottoms, targ)
        t_bc = time.perf_counter() - t0
    finally:
        sys.stdout = _stdout
    per_sweep = t_le / max(1, args.sweeps)
    full = args.full_sweeps or args.sweeps
    cpu = platform.processor() or platform.machine()
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                cpu = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    out = {'net': args.net, 'weights': n_w, 'relations': len(rels), 'sweeps_timed': args.sweeps, 'equalization_s': t_le,
           's_per_sweep': per_sweep, 'bias_correction_s': t_bc, 'full_pass_sweeps': full,
           'value': n_w / (per_sweep * full + t_bc), 'unit': 'weights/s', 'cores': cores, 'kind': 'reference',
           'what': 'unmodified reference (oracle/_ref: dfq._layer_equalization over the relation list as dfq.py:84-108 does, '
                   '{} sweeps timed, {} s/sweep x {} sweeps of a full pass + dfq.bias_correction once), torch {} CPU, {} threads'
                   .format(args.sweeps, round(per_sweep, 3), full, torch.__version__, cores),
           'where': 'this box: {} ({} logical cores), {}'.format(cpu, os.cpu_count(), platform.platform())}
    print(json.dumps(out))
    return 0


if __name__ == '__main__':
    sys.exit(main())
