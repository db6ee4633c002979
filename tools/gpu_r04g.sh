#!/bin/bash
# round 4: resident engine -- strict arrivals made after the sweep's tail; parity + latency (3 runs)
tag=r04g
mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests/test_engine_parity.py tests/test_full_reference.py tests/test_errors.py -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/$tag/pytest.log
for rep in 1 2 3; do timeout 300 python tools/lat.py mobilenet_v2 deeplab_mnv2:60 2>/dev/null | tee -a gpurun_out/$tag/lat.json; done
timeout 300 python tools/trace_resident.py mobilenet_v2 8 > gpurun_out/$tag/trace.txt 2>&1; tail -2 gpurun_out/$tag/trace.txt
grep -E "^layer +(4[6-9]|5[0-2]) x" gpurun_out/$tag/trace.txt | cut -c1-420
