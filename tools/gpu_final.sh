#!/bin/bash
# end-of-round check: GPU suite, smoke(), the default bench line (copied to profiles/ by hand)
mkdir -p gpurun_out/final
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest.log 2>&1; echo "pytest rc=$? ${SECONDS}s"; grep -E "passed|failed" gpurun_out/final/pytest.log | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "bench rc=$?"
python tools/bench_line.py gpurun_out/r03_bench_default.json | cut -c1-250
