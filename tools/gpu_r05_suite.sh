#!/bin/bash
# round 5, first GPU call: the GPU suite as the driver runs it, smoke(), the default bench line, the single-network latency probe
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r05/gpu_suite.txt 2>&1
tail -3 gpurun_out/r05/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err < /dev/null
python tools/bench_line.py gpurun_out/r05/bench_default.json | head -40
timeout 300 python tools/lat.py mobilenet_v2 deeplab_mnv2 deeplab_mnv2:60 resnet18 2>&1 | tail -12
