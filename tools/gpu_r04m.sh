#!/bin/bash
# round 4: resident engine A/B on one box -- strict arrival after the sweep's tail (in-tree) vs right behind the publication (variant)
for rep in 1 2 3; do
for lib in dfq_amd/libdfq_hip.so variants/libdfq_hip_early.so; do
  echo "== $lib"
  DFQ_HIP_LIB=$PWD/$lib timeout 300 python tools/lat.py mobilenet_v2 deeplab_mnv2:60 2>/dev/null
done; done
