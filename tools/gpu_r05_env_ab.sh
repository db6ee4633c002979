#!/bin/bash
# round 5: single-network latency (tools/lat.py) of the in-tree library under a list of environment settings, two rounds.
# usage: tools/gpu_r05_env_ab.sh "ENV1=a ENV2=b" "ENV1=c" ... -- nets...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
sets=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do sets+=("$1"); shift; done
shift
NETS="${@:-mobilenet_v2 deeplab_mnv2:60}"
{
for round in 1 2; do
  for s in "" "${sets[@]}"; do
    echo "== [$s] (round $round)"; env $s timeout 120 python tools/lat.py $NETS 2>/dev/null
  done
done
} > gpurun_out/r05/env_ab.txt 2>&1
cat gpurun_out/r05/env_ab.txt
