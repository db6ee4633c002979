#!/bin/bash
# round 4: streaming engine -- full-row tiles take their own statistics (row_tile local mode): parity, headline A/B over DFQ_LE_LOCAL_ROW
tag=r04i
mkdir -p gpurun_out/$tag
timeout 900 python -m pytest tests/test_engine_parity.py tests/test_full_reference.py tests/test_errors.py -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/$tag/pytest.log
for rep in 1 2; do
for l in 512 0 1024; do
  echo "local_row=$l"
  DFQ_LE_LOCAL_ROW=$l timeout 600 python bench.py --others= --act-shape= --sharded= --distill= --pcie= --cpu-seconds 0 --lazy-steps 0 > gpurun_out/$tag/bench_row$l.json 2> gpurun_out/$tag/bench_row$l.err; echo "bench rc=$?"
  python tools/bench_line.py gpurun_out/$tag/bench_row$l.json | head -1 | cut -c1-300
done; done
