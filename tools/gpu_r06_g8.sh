#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
short="--steps 8 --warmup 3 --cpu-seconds 0 --sharded= --distill= --pcie= --lazy-steps 0 --act-shape= --others="
for round in 1 2; do
for v in g4 g8; do
  unset DFQ_LE_CF_GROUP
  [ $v = g8 ] && export DFQ_LE_CF_GROUP=8
  timeout 300 python bench.py $short > gpurun_out/r06/g_$v$round.json 2> gpurun_out/r06/g_$v$round.err
  python - <<PY
import json
d=json.load(open('gpurun_out/r06/g_$v$round.json'))
r=d['roofline']
print('$v$round', 'value %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'alone', d['config'].get('one_unit_alone_ms'),
      'level us %.1f frac %.3f' % (r['us_per_launch'], r['frac']), 'sweep wall %.1f' % r['sweep_wall_us'], 'fr', (r.get('free_running') or {}).get('us_per_launch'))
PY
done
done
timeout 600 python -m pytest tests/test_range_parity.py tests/test_sharded.py -m gpu -q 2>&1 | tail -3
