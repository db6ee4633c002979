#!/usr/bin/env python
"""The smallest process that launches the timed batch's kernels: ONE batched unit (default 32 MobileNetV2), a few forced
sweeps + bias correction.  For counter passes (tools/pmc_level.sh): rocprofv3 --pmc serialises every dispatch, and bench.py's
set-up alone is 230 000 small copy kernels (deep copies of every step's networks), which made the batch-32 counter pass of
round 1 abort and of round 2 time out.   python tools/pmc_unit.py [--batch 32] [--sweeps 4]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--sweeps', type=int, default=4)
ap.add_argument('--net', default='mobilenet_v2')
ap.add_argument('--restarts', type=int, default=1, help='repeat the restart (bootstrap launch) this many times')
ap.add_argument('--no-lazy', action='store_true', help='without the opt-in lazy-scale engine')
ap.add_argument('--le-only', action='store_true', help='the equalisation launches only (a counter pass over everything aborted inside rocprofv3 at batch 64)')
args = ap.parse_args()
dev = torch.device('cuda', 0)
protos = [bench.prepare(args.net, seed=i, dev=dev) for i in range(args.batch)]
from dfq_amd import dfq
if args.batch == 1:
    le = dfq.build_le_plan(protos[0][1], protos[0][3], bench.TARG)
    bc, _ = dfq.build_bc_plan(protos[0][1], protos[0][2], bench.TARG)
else:
    le = dfq.build_le_plan_batch([(g, r) for (_, g, _, r) in protos], bench.TARG)
    bc = dfq.build_bc_plan_batch([(g, b) for (_, g, b, _) in protos], bench.TARG)
for _ in range(args.restarts - 1):
    le.enqueue(0, restart=True, max_sweeps=args.sweeps, converge_thres=-1.0, converge_count=10 ** 9)
le.enqueue(args.sweeps, restart=True, max_sweeps=args.sweeps, converge_thres=-1.0, converge_count=10 ** 9)
if not args.le_only:
    bc.run()
torch.cuda.synchronize()
if args.batch > 1 and not args.le_only and not args.no_lazy:      # the opt-in lazy-scale engine on a second set of the same networks (its kernels' counters: lz_stats_kernel, rebuild_kernel)
    import copy
    nets = [copy.deepcopy(p) for p in protos]
    lz = dfq.LazyLEPlan([(g, r) for (_, g, _, r) in nets], bench.TARG)
    lz.run(args.sweeps)
    torch.cuda.synchronize()
print('pmc_unit: batch', args.batch, 'sweeps', le.query()['sweeps'], 'workgroups per launch', le.level_info(0)['workgroups'])
