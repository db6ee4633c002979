#!/bin/bash
# quick resident-engine check on the GPU: parity of the engines, latency of the single-network pass, phase trace
tag=${1:-x}
mkdir -p gpurun_out/$tag
timeout 600 python -m pytest tests/test_engine_parity.py -m gpu -x -q -k "resident or full_size or engines_agree" > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/$tag/pytest.log
timeout 300 python tools/lat.py mobilenet_v2 deeplab_mnv2:60 2>/dev/null | tee gpurun_out/$tag/lat.json
timeout 300 python tools/trace_resident.py mobilenet_v2 8 > gpurun_out/$tag/trace.txt 2>&1; tail -1 gpurun_out/$tag/trace.txt
