for ab in 0 4 1; do
  lib=$PWD/dfq_amd/variants/libdfq_hip_ab$ab.so
  [ $ab = 0 ] && lib=$PWD/dfq_amd/libdfq_hip.so
  DFQ_HIP_LIB=$lib timeout 60 python bench.py --batch 8 --streams 1 --steps 2 --warmup 1 --cpu-seconds 0 --sweeps 20 --force-sweeps > gpurun_out/ab_$ab.json 2> gpurun_out/ab_$ab.err
  echo -n "ablate $ab: "; python tools/bench_line.py gpurun_out/ab_$ab.json | cut -d'|' -f2-
done
