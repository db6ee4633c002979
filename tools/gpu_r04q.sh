#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
DFQ_PLAN_TIMING=1 timeout 300 python tools/plan_cost.py 32 2>&1 | grep "dfq\]\|one alloc" | tail -8
timeout 400 python bench.py --others "" --sharded "" --distill "" --pcie "" --lazy-steps 0 --cpu-seconds 0 > gpurun_out/r04q_arena.json 2> gpurun_out/r04q_arena.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04q_arena.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['host_inclusive'], d['config'].get('batch_layout_ms_per_unit'), d['roofline']['us_per_launch'])
PY
