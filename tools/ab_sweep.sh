#!/bin/bash
# A/B of the sweep-launch variants on ONE GPU box (box-to-box variance is larger than most kernel tweaks):
#   new    le_sweep_kernel, library as built          per-tile  le_level_kernel (DFQ_LE_PERSIST=0)
#   w3/w1  variants/libdfq_hip_w{3,1}.so (le_sweep_kernel built for 3 / any number of waves per SIMD)
# usage: tools/ab_sweep.sh [bench flags]
mkdir -p gpurun_out
FLAGS="--steps 6 --warmup 2 --cpu-seconds 0 --others= --act-shape= --sharded= $*"
for round in 1 2; do
  for which in new pertile w3 w1; do
    lib=$PWD/dfq_amd/libdfq_hip.so; persist=1
    [ $which = pertile ] && persist=0
    [ $which = w3 ] && lib=$PWD/variants/libdfq_hip_w3.so
    [ $which = w1 ] && lib=$PWD/variants/libdfq_hip_w1.so
    [ -f $lib ] || continue
    DFQ_LE_PERSIST=$persist DFQ_HIP_LIB=$lib timeout 200 python bench.py $FLAGS > gpurun_out/ab_$which$round.json 2> gpurun_out/ab_$which$round.err < /dev/null
    echo -n "$which$round: "; python tools/bench_line.py gpurun_out/ab_$which$round.json | cut -c1-220
  done
done
