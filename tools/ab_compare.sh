#!/bin/bash
# A/B of two builds of the library on the same GPU box (box-to-box variance is larger than most kernel tweaks):
# usage: tools/ab_compare.sh <variant .so> [bench flags]
V=$1; shift
for round in 1 2; do
  for which in new old; do
    lib=$PWD/dfq_amd/libdfq_hip.so; [ $which = old ] && lib=$V
    DFQ_HIP_LIB=$lib timeout 150 python bench.py --streams 1 --steps 6 --warmup 2 --cpu-seconds 0 "$@" > gpurun_out/ab_$which$round.json 2> gpurun_out/ab_$which$round.err
    echo -n "$which$round: "; python tools/bench_line.py gpurun_out/ab_$which$round.json | cut -c1-200
  done
done
