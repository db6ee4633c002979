#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
short="--steps 8 --warmup 3 --cpu-seconds 0 --sharded= --distill= --pcie= --lazy-steps 0 --act-shape="
timeout 300 python bench.py $short > gpurun_out/r06/bench_t.json 2> gpurun_out/r06/bench_t.err
python - <<PY
import json
d=json.load(open('gpurun_out/r06/bench_t.json'))
r=d['roofline']
print('value %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'alone', d['config'].get('one_unit_alone_ms'), 'single %.4g' % d.get('value_single_network', 0),
      'level us %.1f frac %.3f' % (r['us_per_launch'], r['frac']), 'sweep wall %.1f' % r['sweep_wall_us'], 'fr', r.get('free_running'))
for o in d['config'].get('others', []): print('   ', o['net'], o['ms'], o['equalization_ms'], o['bias_correction_ms'], o['roofline_frac'])
PY
timeout 300 python tools/trace_classes.py 32 > gpurun_out/r06/trace_classes_32.txt 2>&1
head -40 gpurun_out/r06/trace_classes_32.txt
