#!/bin/bash
# round 4: streaming engine -- depthwise rows take their own statistics (LeRelDev::local_r1): parity, headline A/B
tag=r04h
mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests/test_engine_parity.py tests/test_full_reference.py tests/test_errors.py -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/$tag/pytest.log
for rep in 1 2; do
for l in 1 0; do
  echo "local_r1=$l"
  DFQ_LE_LOCAL_R1=$l timeout 600 python bench.py --others= --act-shape= --sharded= --distill= --pcie= --cpu-seconds 0 --lazy-steps 0 > gpurun_out/$tag/bench_local$l.json 2> gpurun_out/$tag/bench_local$l.err; echo "bench rc=$?"
  python tools/bench_line.py gpurun_out/$tag/bench_local$l.json | head -1 | cut -c1-300
done; done
