#!/bin/bash
# small measurement records kept under profiles/: dispatch order litmus, host cost of plan creation, PCIe-inclusive entry points
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${ROUND:-r04}
( for a in "21280 256 200" "100000 256 200" "4096 1024 200" "21280 256 0" "50000 64 50"; do timeout 120 tools/litmus/dispatch_order $a; done ) > gpurun_out/${R}_dispatch_order.txt 2>&1
DFQ_PLAN_TIMING=1 timeout 300 python tools/plan_cost.py 32 2>&1 | grep -E "dfq\]|LE tables|one alloc" | sed -n '1,4p;/LE tables/p;/one alloc/p' | head -16 > gpurun_out/${R}_plan_cost.txt
timeout 300 python tools/pcie_cost.py mobilenet_v2 4 2>/dev/null | grep -E "rep [123]" > gpurun_out/${R}_pcie_cost.txt
tail -4 gpurun_out/${R}_dispatch_order.txt; tail -3 gpurun_out/${R}_plan_cost.txt; tail -2 gpurun_out/${R}_pcie_cost.txt | cut -c1-300
