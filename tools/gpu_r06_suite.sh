#!/bin/bash
# round 6: the GPU suite as the driver runs it + the bench line's sharded entries
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06/gpu_suite_full.txt 2>&1
echo "suite ${SECONDS}s"; tail -6 gpurun_out/r06/gpu_suite_full.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py --steps 6 --warmup 2 --cpu-seconds 0 --distill= --pcie= --lazy-steps 0 --act-shape= --others= > gpurun_out/r06/bench_sh.json 2> gpurun_out/r06/bench_sh.err
python - <<PY
import json
d=json.load(open('gpurun_out/r06/bench_sh.json'))
print('value %.4g' % d['value'], 'single %.4g' % d.get('value_single_network', 0), d.get('latency'))
for s in d.get('sharded', []): print({k: v for k, v in s.items() if k != 'what'})
PY
