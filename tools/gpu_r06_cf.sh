#!/bin/bash
# round 6: free-running segments of the streaming engine -- GPU suite, then the bench A/B over the group depth
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06/gpu_suite.txt 2>&1
tail -4 gpurun_out/r06/gpu_suite.txt
short="--steps 8 --warmup 3 --cpu-seconds 0 --sharded= --distill= --pcie= --lazy-steps 0 --act-shape="
for v in cf4 cf0 cf8 cf2 cf4b; do
  case $v in
    cf0) export DFQ_LE_CF=0; unset DFQ_LE_CF_GROUP;;
    cf8) unset DFQ_LE_CF; export DFQ_LE_CF_GROUP=8;;
    cf2) unset DFQ_LE_CF; export DFQ_LE_CF_GROUP=2;;
    *) unset DFQ_LE_CF; unset DFQ_LE_CF_GROUP;;
  esac
  timeout 300 python bench.py $short > gpurun_out/r06/bench_$v.json 2> gpurun_out/r06/bench_$v.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r06/bench_$v.json'))
    r=d['roofline']
    print('$v', 'value %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'alone', d['config'].get('one_unit_alone_ms'), 'single %.4g' % d.get('value_single_network', 0),
          'level us %.1f frac %.3f' % (r['us_per_launch'], r['frac']), 'sweep wall %.1f' % r['sweep_wall_us'], 'fr', r.get('free_running'))
    for o in d['config'].get('others', []): print('   ', o['net'], o['ms'], o['equalization_ms'], o['bias_correction_ms'], o['roofline_frac'])
except Exception as e:
    print('$v failed', e); print(open('gpurun_out/r06/bench_$v.err').read()[-1500:])
PY
done
