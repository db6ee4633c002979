#!/bin/bash
# round 4: folded depthwise steps of the correction chain -- parity, single-network latency fold on / off, batch headline
tag=r04b
mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests/test_engine_parity.py tests/test_full_reference.py -m gpu -x -q -k "bias or bc or folded or full or pipeline" > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/$tag/pytest.log
echo "fold on"; timeout 300 python tools/lat.py mobilenet_v2 resnet18 deeplab_mnv2:60 2>/dev/null | tee gpurun_out/$tag/lat_fold.json
echo "fold off"; DFQ_BC_FOLD=0 timeout 300 python tools/lat.py mobilenet_v2 resnet18 deeplab_mnv2:60 2>/dev/null | tee gpurun_out/$tag/lat_nofold.json
timeout 600 python bench.py --others= --act-shape= --sharded= --distill= --pcie= --cpu-seconds 0 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; echo "bench rc=$?"
python tools/bench_line.py gpurun_out/$tag/bench.json | cut -c1-300
DFQ_BC_FOLD=0 timeout 600 python bench.py --others= --act-shape= --sharded= --distill= --pcie= --cpu-seconds 0 > gpurun_out/$tag/bench_nofold.json 2> gpurun_out/$tag/bench_nofold.err; echo "bench rc=$?"
python tools/bench_line.py gpurun_out/$tag/bench_nofold.json | cut -c1-300
