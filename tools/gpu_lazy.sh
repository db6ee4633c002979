#!/bin/bash
mkdir -p gpurun_out/lazy
timeout 900 python -m pytest tests/test_engine_parity.py tests/test_full_reference.py -m gpu -x -q -k "lazy or stage_wise" -s > gpurun_out/lazy/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|worst" gpurun_out/lazy/pytest.log | tail -5
timeout 600 python bench.py --others= --act-shape= --sharded= --cpu-seconds 0 --distill= --pcie= > gpurun_out/lazy/bench.json 2> gpurun_out/lazy/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/lazy/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/lazy/bench.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'])
lz=d['lazy_scale']; print('lazy', lz['value'], lz['ms_per_step'], lz['equalization_ms'], lz['roofline']['frac'], lz['roofline']['us_per_sweep'], lz['launches_per_pass'])
PY
