#!/bin/bash
# where the C side of plan creation goes: plain timing, then HIP API statistics of the same script
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python tools/plan_cost.py 32 2>&1 | tail -5
rm -rf gpurun_out/prof_plan
timeout 300 rocprofv3 --hip-trace --stats --output-format csv -d gpurun_out/prof_plan -o plan -- python tools/plan_cost.py 32 > /dev/null 2>&1
S=$(find gpurun_out/prof_plan -name "*hip_api_stats.csv" | head -1)
[ -n "$S" ] && head -25 "$S" | cut -c1-160
find gpurun_out/prof_plan -name "*trace.csv" -delete
