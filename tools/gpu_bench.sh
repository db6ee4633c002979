#!/bin/bash
tag=${1:-bench}
mkdir -p gpurun_out/$tag
SECONDS=0; timeout 900 python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; echo "bench rc=$? wall ${SECONDS}s"; tail -2 gpurun_out/$tag/bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/$tag/bench.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'single', d.get('value_single_network'), 'lat', d['latency']['single_network_pass_ms'], d['latency']['equalization_gpu_ms'], d['latency']['bias_correction_gpu_ms'])
print('alone', d['config']['one_unit_alone_ms'], 'plan_build', d['config']['plan_build_ms_per_unit'])
print('roofline', d['roofline']['frac'], d['roofline']['us_per_launch'])
for o in d['config'].get('others', []): print(o['net'], o['ms'], o['equalization_ms'], o['bias_correction_ms'], o['roofline_frac'])
dr=d['config'].get('distill_range'); print('distill', {k: dr[k] for k in dr if k not in ('what',)} if dr else None)
print('pcie', d.get('pcie_inclusive'))
print('sharded', {k: v for k, v in d.get('sharded', {}).items() if k != 'what'})
PY
