#!/bin/bash
mkdir -p gpurun_out/pcie
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pcie/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/pcie/pytest.log | tail -2
python - <<'PY' 2>/dev/null
import bench, torch, json
dev = bench._device(0)
for i in range(4):
    r = bench.pcie_inclusive_pass('mobilenet_v2', reps=1)
    print(i, {k: round(v, 2) for k, v in r.items() if k.endswith('_ms')})
d = bench.distill_range_pass('mobilenet_v2', [64, 3, 224, 224], 2, dev)
print('after distill')
for i in range(3):
    r = bench.pcie_inclusive_pass('mobilenet_v2', reps=1)
    print(i, {k: round(v, 2) for k, v in r.items() if k.endswith('_ms')})
PY
