#!/bin/bash
tag=${1:-bc}
mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests/test_engine_parity.py tests/test_full_reference.py -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/$tag/pytest.log
timeout 300 python tools/lat.py mobilenet_v2 resnet18 deeplab_mnv2:60 2>/dev/null | tee gpurun_out/$tag/lat.json
timeout 600 python bench.py --others= --act-shape= --sharded= --cpu-seconds 0 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; echo "bench rc=$?"
python tools/bench_line.py gpurun_out/$tag/bench.json | cut -c1-300
