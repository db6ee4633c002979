#!/bin/bash
# round 4: host-side cost of building a batch's two plans (Python tables vs C-side plan creation)
tag=r04j
mkdir -p gpurun_out/$tag
timeout 600 python tools/plan_cost.py 32 2>&1 | tail -5 | tee gpurun_out/$tag/plan_cost.txt
timeout 600 python -X importtime -c "pass" 2>/dev/null; 
timeout 900 python - <<'PY' 2>&1 | tail -45 | tee gpurun_out/r04j/profile.txt
import copy, cProfile, pstats, sys, os
sys.path.insert(0, os.getcwd())
import torch, torch.nn as nn
import bench
from dfq_amd import _ffi, dfq
dev = torch.device('cuda', 0)
protos = [bench.prepare('mobilenet_v2', seed=i, dev=dev) for i in range(32)]
nets = [copy.deepcopy(p) for p in protos]
stage = _ffi.Stage()
for rep in range(2):
    nets = [copy.deepcopy(p) for p in protos]
    pr = cProfile.Profile(); pr.enable()
    lt = dfq._fast_le_tables([(g, r) for (_, g, _, r) in nets], bench.TARG, stage.device)
    le = dfq.LEPlan(lt, None, stage=stage)
    bt = dfq._fast_bc_tables([(g, b) for (_, g, b, _) in nets], bench.TARG, nn.BatchNorm2d, stage.device)
    bc = dfq.BCPlan(bt, None, stage=stage)
    pr.disable()
    le.close(); bc.close()
pstats.Stats(pr).sort_stats('tottime').print_stats(25)
PY
