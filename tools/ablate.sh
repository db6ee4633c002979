#!/bin/bash
# tuning aid: per-level time of the equalisation sweep with parts of the tile kernels compiled out
# (variant libraries built with -DDFQ_LE_ABLATE=bits into dfq_amd/variants/, selected through DFQ_HIP_LIB)
B=${1:-8}
for ab in 0 1 2 4 8 15; do
  lib=$PWD/dfq_amd/variants/libdfq_hip_ab$ab.so
  [ $ab = 0 ] && lib=$PWD/dfq_amd/libdfq_hip.so
  DFQ_HIP_LIB=$lib timeout 120 python bench.py --batch $B --streams 1 --steps 2 --warmup 1 --cpu-seconds 0 --sweeps 47 --force-sweeps > gpurun_out/ab_$ab.json 2> gpurun_out/ab_$ab.err
  echo -n "ablate $ab: "; python tools/bench_line.py gpurun_out/ab_$ab.json | cut -d'|' -f2-
done
