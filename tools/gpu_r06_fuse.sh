#!/bin/bash
# round 6: layers scaled along both axes read once per sweep (slab tiles, DFQ_LE_FUSE) -- parity suite, then A/B at batch 32
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_engine_parity.py tests/test_arena.py tests/test_errors.py -m gpu -x -q > gpurun_out/r06/gpu_fuse_suite.txt 2>&1
tail -3 gpurun_out/r06/gpu_fuse_suite.txt
short="--steps 8 --warmup 3 --cpu-seconds 0 --sharded= --distill= --pcie= --lazy-steps 0 --act-shape= --others="
for round in 1 2; do
for v in fuse1 fuse0; do
  case $v in
    fuse1) export DFQ_LE_FUSE=1;;
    fuse0) export DFQ_LE_FUSE=0;;
  esac
  timeout 300 python bench.py $short > gpurun_out/r06/f_$v$round.json 2> gpurun_out/r06/f_$v$round.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/r06/f_$v$round.json'))
    r=d['roofline']
    print('$v$round', 'value %.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'alone', d['config'].get('one_unit_alone_ms'),
          'level us %.1f frac %.3f bytes %.0f' % (r['us_per_launch'], r['frac'], r['bytes_per_launch']), 'sweep wall %.1f' % r['sweep_wall_us'], 'all GBps %.0f' % r['GBps_per_sweep_all_kernels'], 'fr', (r.get('free_running') or {}).get('us_per_launch'))
except Exception as e:
    print('$v$round failed', e); print(open('gpurun_out/r06/f_$v$round.err').read()[-1500:])
PY
done
done
unset DFQ_LE_FUSE
timeout 300 python tools/trace_classes.py 32 > gpurun_out/r06/trace_classes_32_fused.txt 2>&1
head -24 gpurun_out/r06/trace_classes_32_fused.txt
