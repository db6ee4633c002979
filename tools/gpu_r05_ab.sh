#!/bin/bash
# round 5: single-network latency (tools/lat.py) of the in-tree library against every variants/libdfq_hip_*.so, two rounds;
# usage: tools/gpu_r05_ab.sh [nets...]     (environment switches of the library apply to all of them)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
NETS="${@:-mobilenet_v2 deeplab_mnv2:60}"
{
for round in 1 2; do
  for lib in dfq_amd/libdfq_hip.so variants/libdfq_hip_*.so; do
    [ -f $lib ] || continue
    echo "== $lib (round $round)"; DFQ_HIP_LIB=$PWD/$lib timeout 120 python tools/lat.py $NETS 2>/dev/null
  done
  echo "== in-tree, DFQ_COOPERATIVE=0 (round $round)"; DFQ_COOPERATIVE=0 timeout 120 python tools/lat.py $NETS 2>/dev/null
done
} > gpurun_out/r05/ab.txt 2>&1
cat gpurun_out/r05/ab.txt
