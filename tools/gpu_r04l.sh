#!/bin/bash
# round 4: BC with two launches (min/max + cache refresh; chain) -- parity + latency; then the PMC passes of the final sweep kernel
tag=r04l
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_engine_parity.py tests/test_full_reference.py tests/test_errors.py -m gpu -x -q -k "bias or bc or folded or pipeline or abandoned or full" > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/$tag/pytest.log | tail -1
for rep in 1 2; do timeout 300 python tools/lat.py mobilenet_v2 resnet18 deeplab_mnv2:60 2>/dev/null | tee -a gpurun_out/$tag/lat.json; done
timeout 600 python bench.py --others= --act-shape= --sharded= --distill= --pcie= --cpu-seconds 0 --lazy-steps 0 > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; echo "bench rc=$?"
python tools/bench_line.py gpurun_out/$tag/bench.json | head -1 | cut -c1-300
PMC_TIMEOUT=400 bash tools/pmc_level.sh --batch 32 --sweeps 4 < /dev/null
ls -la gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE | head -12
