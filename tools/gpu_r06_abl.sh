#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
short="--steps 6 --warmup 2 --cpu-seconds 0 --sharded= --distill= --pcie= --lazy-steps 0 --act-shape= --others= --sweeps 47 --force-sweeps"
for round in 1 2; do
for v in base abl16; do
  unset DFQ_HIP_LIB
  [ $v = abl16 ] && export DFQ_HIP_LIB=$PWD/variants/libdfq_hip_abl16.so
  timeout 300 python bench.py $short > gpurun_out/r06/a_$v$round.json 2> gpurun_out/r06/a_$v$round.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r06/a_$v$round.json'))
    r=d['roofline']
    print('$v$round', 'value %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'level us %.1f' % r['us_per_launch'], 'sweep wall %.1f' % r['sweep_wall_us'])
except Exception as e:
    print('$v$round failed', e); print(open('gpurun_out/r06/a_$v$round.err').read()[-800:])
PY
done
done
