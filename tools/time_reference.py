#!/usr/bin/env python
"""Time the UNMODIFIED reference CPU path (SURVEY 8d "CPU baseline timing") in the build container and commit the figure.

    PYTHONDONTWRITEBYTECODE=1 python tools/time_reference.py [--net mobilenet_v2] [--sweeps K]

/root/reference exists only in the build container, never on the GPU box, so bench.py cannot time the reference
itself there: it carries this file's record (profiles/r02_reference_cpu.json) in `cpu_baseline.reference`, labelled with
where it was measured, next to the port it times live.  What is timed: `dfq.cross_layer_equalization` (dfq.py:78-117, the
whole data-dependent loop, or K sweeps of the `_layer_equalization` driver of dfq.py:85-101 with --sweeps) and
`dfq.bias_correction` (dfq.py:173-293) on the synthetic network of bench.py (seed 0), BN folded by the reference's own
merge_batchnorm, torch.set_num_threads(os.cpu_count()).
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
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(1, REF)
sys.dont_write_bytecode = True

import torch                      # noqa: E402
import torch.nn as nn             # noqa: E402

import dfq as ref_dfq                          # noqa: E402  (reference)
from utils import layer_transform as ref_lt    # noqa: E402  (reference)
from utils import relation as ref_rel          # noqa: E402  (reference)
from dfq_amd import synthetic                  # noqa: E402

TARG = [nn.Conv2d, nn.Linear]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--net', default='mobilenet_v2')
    ap.add_argument('--sweeps', type=int, default=0, help='0 = the reference\'s own convergence loop')
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r02_reference_cpu.json'))
    args = ap.parse_args()
    cores = os.cpu_count()
    torch.set_num_threads(cores)
    model, graph, bottoms = synthetic.build(args.net, seed=0)
    ref_lt.merge_batchnorm(model, graph, bottoms, TARG)
    rels = ref_rel.create_relation(graph, bottoms, TARG, delete_single=False)
    n_w = sum(m.weight.numel() for m in graph.values() if type(m) in TARG)
    counter = {'n': 0}
    orig = copy.deepcopy

    def counting(x, *a, **k):                   # the reference deep-copies the graph once per sweep (dfq.py:84)
        if isinstance(x, dict) and 'Data' in x:
            counter['n'] += 1
        return orig(x, *a, **k)
    t0 = time.perf_counter()
    if args.sweeps > 0:
        with torch.no_grad():
            for _ in range(args.sweeps):
                for rr in rels:
                    lf, ls, bn = rr.get_idxs()
                    if graph[lf].bias is None:
                        graph[lf].bias = nn.Parameter(torch.zeros(graph[lf].weight.size(0)), requires_grad=False)
                    graph[lf].weight, graph[ls].weight, graph[lf].bias, S = ref_dfq._layer_equalization(
                        graph[lf].weight, graph[ls].weight, graph[lf].bias, graph[bn].fake_weight, graph[bn].fake_bias)
                    rr.set_scale_vec(S)
        sweeps = args.sweeps
    else:
        copy.deepcopy = counting
        try:
            ref_dfq.cross_layer_equalization(graph, rels, TARG, converge_thres=2e-7)
        finally:
            copy.deepcopy = orig
        sweeps = counter['n']
    t_le = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref_dfq.bias_correction(graph, bottoms, TARG)
    t_bc = time.perf_counter() - t0
    cpu = ''
    try:
        cpu = [l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')][0]
    except Exception:
        pass
    rec = {
        'net': args.net, 'weights': n_w, 'relations': len(rels), 'sweeps': sweeps,
        'equalization_s': t_le, 'bias_correction_s': t_bc, 's_per_sweep': t_le / max(sweeps, 1),
        'value': n_w / (t_le + t_bc), 'unit': 'weights/s', 'cores': cores,
        'what': 'unmodified /root/reference dfq.py:78-117 + dfq.py:173-293 (torch {} CPU, {} threads)'.format(torch.__version__, cores),
        'where': 'build container: {} ({} logical cores), {}'.format(cpu or platform.processor(), cores, platform.platform()),
    }
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    if os.path.exists(args.out):
        allrec = json.load(open(args.out))
    else:
        allrec = {}
    allrec[args.net if args.sweeps == 0 else '{}@{}sweeps'.format(args.net, args.sweeps)] = rec
    json.dump(allrec, open(args.out, 'w'), indent=1)
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
