#!/bin/bash
# round 4: GPU suite (log kept), BC without clear / cache-init launches (latency), PMC passes of the final sweep kernel
tag=r04k
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/$tag/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for rep in 1 2; do timeout 300 python tools/lat.py mobilenet_v2 resnet18 deeplab_mnv2:60 2>/dev/null | tee -a gpurun_out/$tag/lat.json; done
timeout 600 python bench.py --others= --act-shape= --sharded= --distill= --pcie= --cpu-seconds 0 --lazy-steps 0 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; echo "bench rc=$?"
python tools/bench_line.py gpurun_out/$tag/bench.json | head -1 | cut -c1-300
PMC_TIMEOUT=250 bash tools/pmc_level.sh --batch 32 --sweeps 4 < /dev/null
timeout 120 python tools/pmc_digest.py gpurun_out gpurun_out/r04_bench_under_rocprof.json gpurun_out/r04_pmc_summary.json < /dev/null | tail -8
