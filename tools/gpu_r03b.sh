#!/bin/bash
mkdir -p gpurun_out/r03b
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03b/pytest.log 2>&1; echo "pytest rc=$? ${SECONDS}s"; tail -2 gpurun_out/r03b/pytest.log
timeout 300 python tools/lat.py mobilenet_v2 resnet18 deeplab_mnv2:60 2>/dev/null | tee gpurun_out/r03b/lat.json
for w in 8192 4096 2048; do echo "boot work $w"; DFQ_LE_BOOT_WORK=$w timeout 300 python tools/lat.py resnet18 2>/dev/null; done
