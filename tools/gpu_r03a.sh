#!/bin/bash
# round 3, call A: GPU parity suite, baseline bench line, resident tile-shape A/B
mkdir -p gpurun_out/r03a
cd $GRAFT_REPO_ROOT 2>/dev/null || true
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03a/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r03a/pytest.log
timeout 600 python bench.py > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err; echo "bench rc=$?"
python tools/bench_line.py gpurun_out/r03a/bench.json | cut -c1-400
for v in 0 1; do
  echo "== DFQ_RES_EXACT_GROUPS=$v"
  DFQ_RES_EXACT_GROUPS=$v timeout 300 python tools/lat.py mobilenet_v2 deeplab_mnv2:60 2>/dev/null | tee gpurun_out/r03a/lat_exact$v.json
  DFQ_RES_EXACT_GROUPS=$v timeout 300 python tools/trace_resident.py mobilenet_v2 8 > gpurun_out/r03a/trace_exact$v.txt 2>&1
  tail -1 gpurun_out/r03a/trace_exact$v.txt
done
