#!/bin/bash
# bias-correction chain of the batch under the three hand-over protocols
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "" "DFQ_BC_TAGGED=0" "DFQ_BC_MERGED=0" "DFQ_BC_FOLD=0"; do
env $v timeout 300 python bench.py --steps 4 --warmup 1 --others "" --sharded "" --distill "" --pcie "" --lazy-steps 0 --cpu-seconds 0 --no-roofline --act-shape "" > gpurun_out/r04w.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r04w.json').read().strip().splitlines()[-1])
print('[$v]', 'ms/step %.3f' % d['ms_per_step'], 'batch BC %.3f' % d['config']['one_unit_alone_ms']['bias_correction'], 'single BC %.4f' % d['latency']['bias_correction_gpu_ms'])
PY
done
