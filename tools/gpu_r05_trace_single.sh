#!/bin/bash
# kernel trace of ONE network's pass (tools/lat.py) under rocprofv3: the last pass's launches with start / duration, per network
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for NET in ${NETS:-resnet18 mobilenet_v2}; do
  rm -rf gpurun_out/r05/trace_$NET
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r05/trace_$NET -o t -- python tools/lat.py $NET > gpurun_out/r05/trace_$NET.log 2>&1 < /dev/null
  tail -1 gpurun_out/r05/trace_$NET.log
  T=$(find gpurun_out/r05/trace_$NET -name "*kernel_trace.csv" | head -1)
  [ -n "$T" ] && python tools/trace_tail.py "$T" ${TAIL:-26} | tee gpurun_out/r05/trace_$NET.txt
  [ -n "$T" ] && rm -f "$T"
done
