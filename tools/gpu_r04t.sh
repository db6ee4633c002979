#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for b in 128 256 512 1024 2048; do
DFQ_BC_BLOCKS=$b timeout 300 python bench.py --steps 4 --warmup 1 --others "" --sharded "" --distill "" --pcie "" --lazy-steps 0 --cpu-seconds 0 --no-roofline --act-shape "" > gpurun_out/r04t_$b.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r04t_$b.json').read().strip().splitlines()[-1])
print('DFQ_BC_BLOCKS=$b', 'ms/step %.3f' % d['ms_per_step'], d['config']['one_unit_alone_ms'])
PY
done
