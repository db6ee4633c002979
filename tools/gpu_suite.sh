#!/bin/bash
# the GPU suite as the driver runs it (sequential), verdict into gpurun_out/gpu_suite.txt; then the sweep loop's kernel durations
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.txt 2>&1
tail -3 gpurun_out/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/gpu_sweep_gaps.sh 2>&1 | tail -4
