#!/bin/bash
# round 4: resident engine with speculation past the verdict -- parity, latency at several depths / checkpoint periods, trace
tag=r04c
mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests/test_engine_parity.py tests/test_full_reference.py tests/test_errors.py -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/$tag/pytest.log
for cfg in "2 4" "0 4" "1 4" "3 4" "4 4" "2 2" "2 8" "4 8" "1 1"; do
  set -- $cfg
  echo "spec=$1 ckpt=$2"; DFQ_RES_SPEC=$1 DFQ_RES_CKPT=$2 timeout 300 python tools/lat.py mobilenet_v2 deeplab_mnv2:60 2>/dev/null | tee -a gpurun_out/$tag/lat_$1_$2.json
done
timeout 300 python tools/trace_resident.py mobilenet_v2 8 > gpurun_out/$tag/trace.txt 2>&1; tail -2 gpurun_out/$tag/trace.txt
