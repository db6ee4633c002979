#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_gaps
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gaps -o g -- python bench.py --streams 1 --steps 4 --warmup 2 --cpu-seconds 0 --sharded "" --distill "" --pcie "" --others "" --lazy-steps 0 --no-roofline > /dev/null 2>&1
T=$(find gpurun_out/prof_gaps -name "*kernel_trace.csv" | head -1)
python tools/trace_gaps.py "$T"
rm -rf gpurun_out/prof_gaps
