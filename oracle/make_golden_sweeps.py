#!/usr/bin/env python
"""TEST INFRASTRUCTURE: sweep counts of the UNMODIFIED reference loop (dfq.py:78-117) on the benchmark's synthetic
MobileNetV2 for several seeds -> tests/golden/full_sweeps.json.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_sweeps.py [--seeds 0 1 2 ...]

Why: the loop stops when the float32 mean of |W - W_prev| summed over the layers falls to 2e-7 -- a value made of rounding
noise by then -- and the oracle / the engine form that mean as a float64 sum rounded to float32 once, where torch's float32
mean has an unspecified summation order.  The benchmark batch is 32 seeds; this pins the stopping point
of a handful of them against the reference itself (about 75 s of CPU per seed), not just seed 0.
The reference is imported from /root/reference, never copied; nothing on the GPU box reads this script.
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys

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
OUT = os.path.join(ROOT, 'tests', 'golden', 'full_sweeps.json')


def ref_sweeps(net, seed):
    model, graph, bottoms = synthetic.build(net, seed=seed)
    ref_lt.merge_batchnorm(model, graph, bottoms, TARG)
    rels = ref_rel.create_relation(graph, bottoms, TARG, delete_single=False)
    counter = {'n': 0}
    orig = copy.deepcopy

    def counting(x, *a, **k):                   # the reference deep-copies the graph once per sweep (dfq.py:84)
        if isinstance(x, dict) and 'Data' in x:
            counter['n'] += 1
        return orig(x, *a, **k)
    copy.deepcopy = counting
    try:
        ref_dfq.cross_layer_equalization(graph, rels, TARG, converge_thres=2e-7)
    finally:
        copy.deepcopy = orig
    return counter['n']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--net', default='mobilenet_v2')
    ap.add_argument('--seeds', type=int, nargs='*', default=[0, 1, 2, 3, 4, 5, 6, 7])
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    rec = json.load(open(OUT)) if os.path.exists(OUT) else {}
    rec.setdefault(args.net, {})
    for seed in args.seeds:
        n = ref_sweeps(args.net, seed)
        rec[args.net][str(seed)] = n
        print(args.net, 'seed', seed, 'reference sweeps', n, flush=True)
        json.dump(rec, open(OUT, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
