#!/bin/bash
# round 5: what a position of the one-launch correction chain is made of -- the in-tree library against variants built with
# -DDFQ_BC_ABLATE=<bits> (dfq_bc.hip; results wrong by construction) and -DDFQ_BC_F32_MOMENT=1; bc_ms of tools/lat.py, two rounds
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
NETS="${@:-mobilenet_v2 resnet18}"
{
for round in 1 2; do
  for lib in dfq_amd/libdfq_hip.so variants/libdfq_hip_bc*.so; do
    [ -f $lib ] || continue
    echo "== $lib (round $round)"; DFQ_HIP_LIB=$PWD/$lib timeout 120 python tools/lat.py $NETS 2>/dev/null
  done
done
} > gpurun_out/r05/bc_ablate.txt 2>&1
cat gpurun_out/r05/bc_ablate.txt
