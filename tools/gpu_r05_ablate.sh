#!/bin/bash
# round 5: what is a resident sweep's time made of?  Library variants with work compiled OUT of the sweep (DFQ_RES_ABLATE, results wrong,
# pinned sweep counts), the cooperative-launch switch, and the shader clock a short kernel really runs at.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
tools/litmus/clock_rate > gpurun_out/r05/clock_rate.txt 2>&1; cat gpurun_out/r05/clock_rate.txt
NETS="mobilenet_v2:47 deeplab_mnv2:60"
{
echo "== base"; timeout 120 python tools/lat.py $NETS 2>/dev/null
echo "== base, DFQ_COOPERATIVE=0"; DFQ_COOPERATIVE=0 timeout 120 python tools/lat.py $NETS 2>/dev/null
for lib in variants/libdfq_hip_abl*.so; do
  echo "== $lib"; DFQ_HIP_LIB=$PWD/$lib timeout 120 python tools/lat.py $NETS 2>/dev/null
done
echo "== base again"; timeout 120 python tools/lat.py $NETS 2>/dev/null
} > gpurun_out/r05/ablate.txt 2>&1
cat gpurun_out/r05/ablate.txt
