#!/bin/bash
# round 5: is the reducer (one verdict per sweep) what paces the resident loop?  Speculation depths up to "never wait for a verdict".
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
{
for sc in "2 8" "4 8" "8 8" "16 16" "32 32" "64 64"; do
  set -- $sc
  echo "== DFQ_RES_SPEC=$1 DFQ_RES_CKPT=$2"; DFQ_RES_SPEC=$1 DFQ_RES_CKPT=$2 timeout 120 python tools/lat.py mobilenet_v2:47 deeplab_mnv2:60 2>&1 | grep '^{'
done
} > gpurun_out/r05/spec.txt 2>&1
cat gpurun_out/r05/spec.txt
