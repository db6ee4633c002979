#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 300 python bench.py --others "" --sharded "" --distill "" --pcie "" --lazy-steps 0 --cpu-seconds 0 --no-roofline --act-shape "" > gpurun_out/r04z_$i.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r04z_$i.json').read().strip().splitlines()[-1])
print('run $i', 'value %.4e' % d['value'], 'ms/step %.3f' % d['ms_per_step'], d['config']['one_unit_alone_ms'])
PY
done
