#!/bin/bash
# round 6: lean tiles woven into the sweep's launch (default build) against their own launch (env switch; the build without the
# woven path in le_level_kernel at all), two rounds
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
short="--steps 8 --warmup 3 --cpu-seconds 0 --sharded= --distill= --pcie= --lazy-steps 0 --act-shape= --others="
for round in 1 2; do
for v in weave envoff noweave; do
  unset DFQ_HIP_LIB DFQ_LE_CF_WEAVE
  case $v in
    envoff) export DFQ_LE_CF_WEAVE=0;;
    noweave) export DFQ_HIP_LIB=$PWD/variants/libdfq_hip_noweave.so;;
  esac
  timeout 300 python bench.py $short > gpurun_out/r06/w_$v$round.json 2> gpurun_out/r06/w_$v$round.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r06/w_$v$round.json'))
    r=d['roofline']
    print('$v$round', 'value %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'alone', d['config'].get('one_unit_alone_ms'),
          'level us %.1f frac %.3f' % (r['us_per_launch'], r['frac']), 'sweep wall %.1f' % r['sweep_wall_us'], 'all GBps %.0f' % r['GBps_per_sweep_all_kernels'], 'fr', (r.get('free_running') or {}).get('us_per_launch'))
except Exception as e:
    print('$v$round failed', e); print(open('gpurun_out/r06/w_$v$round.err').read()[-1500:])
PY
done
done
