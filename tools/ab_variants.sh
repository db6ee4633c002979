#!/bin/bash
# A/B of library builds on ONE GPU box: the in-tree library against every variants/libdfq_hip_*.so, two rounds.
# usage: tools/ab_variants.sh [bench flags]
mkdir -p gpurun_out
FLAGS="--cpu-seconds 0 --others= --act-shape= --sharded= --lazy-steps 0 --pcie= --distill= $*"
for round in 1 2; do
  for lib in dfq_amd/libdfq_hip.so variants/libdfq_hip_*.so; do
    [ -f $lib ] || continue
    tag=$(basename $lib .so | sed 's/libdfq_hip_\?//'); [ -z "$tag" ] && tag=base
    DFQ_HIP_LIB=$PWD/$lib timeout 200 python bench.py $FLAGS > gpurun_out/ab_$tag$round.json 2> gpurun_out/ab_$tag$round.err < /dev/null
    echo -n "$tag$round: "; python tools/bench_line.py gpurun_out/ab_$tag$round.json | head -1 | cut -c1-220
  done
done
