#!/bin/bash
# One GPU-box call at the end of a change: the GPU suite as the driver runs it, smoke(), a short bench line.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gpu_suite.txt 2>&1
echo "suite ${SECONDS}s"; grep -E "passed|failed" gpurun_out/gpu_suite.txt | tail -2; grep "^FAILED" gpurun_out/gpu_suite.txt | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py --steps 6 --warmup 2 --cpu-seconds 0 --lazy-steps 0 > gpurun_out/bench_round.json 2> gpurun_out/bench_round.err
python tools/bench_line.py gpurun_out/bench_round.json
