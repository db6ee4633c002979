#!/bin/bash
# deferred stores of the streaming engine on the GPU: parity tests, then the headline bench at depths 1 / 2 / 4
tag=${1:-defer}
mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests/test_engine_parity.py tests/test_full_reference.py -m gpu -x -q -k "deferred or full_size or batched or launch_modes or heterogeneous" > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/$tag/pytest.log
for d in 2 1 4 2; do
  DFQ_LE_DEFER=$d timeout 600 python bench.py --cpu-seconds 0 --others '' --lazy-steps 0 --pcie '' --distill '' --sharded '' --act-shape '' > gpurun_out/$tag/bench_d$d.json 2> gpurun_out/$tag/bench_d$d.err; echo "depth $d rc=$?"
  python - <<PY
import json
d=json.load(open('gpurun_out/$tag/bench_d$d.json'))
r=d['roofline']
print('depth', r.get('deferred_store_depth'), 'value', d['value'], 'ms/step', d['ms_per_step'], 'us/launch', r['us_per_launch'], 'frac', r['frac'], 'bytes', r['bytes_per_launch'], 'eq every-sweep GB/s', r.get('equivalent_GBps_storing_every_sweep'))
PY
done
