#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_arena.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/plan_cost.py 32 2>&1 | tail -10
timeout 400 python bench.py --others "" --sharded "" --distill "" --pcie "" --lazy-steps 0 --cpu-seconds 0 > gpurun_out/r04p_arena.json 2> gpurun_out/r04p_arena.err
python tools/bench_line.py gpurun_out/r04p_arena.json
DFQ_BENCH_ARENA=0 timeout 400 python bench.py --others "" --sharded "" --distill "" --pcie "" --lazy-steps 0 --cpu-seconds 0 > gpurun_out/r04p_scattered.json 2> gpurun_out/r04p_scattered.err
python tools/bench_line.py gpurun_out/r04p_scattered.json
python - <<'PY'
import json
for n in ('arena','scattered'):
    d=json.loads(open('gpurun_out/r04p_%s.json'%n).read().strip().splitlines()[-1])
    print(n, d['value'], d['ms_per_step'], d['host_inclusive'], d['config'].get('batch_layout_ms_per_unit'), d['roofline']['us_per_launch'])
PY
