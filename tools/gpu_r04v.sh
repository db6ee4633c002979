#!/bin/bash
# bias-correction chain: pause between polls of a tagged slot (compile-time cap of the back-off), batch of 32 and one network
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in base pc8 pc32 pc128; do
if [ $v = base ]; then unset DFQ_HIP_LIB; else export DFQ_HIP_LIB=$GRAFT_REPO_ROOT/variants/libdfq_hip_$v.so; fi
timeout 300 python bench.py --steps 6 --warmup 2 --others "resnet18" --sharded "" --distill "" --pcie "" --lazy-steps 0 --cpu-seconds 0 --no-roofline --act-shape "" > gpurun_out/r04v_$v.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r04v_$v.json').read().strip().splitlines()[-1])
print('$v', 'ms/step %.3f' % d['ms_per_step'], 'batch BC %.3f' % d['config']['one_unit_alone_ms']['bias_correction'], 'single BC %.4f' % d['latency']['bias_correction_gpu_ms'], 'resnet BC %.4f' % d['config']['others'][0]['bias_correction_ms'])
PY
done
