#!/bin/bash
# like ab_env.sh but with the default two streams (what the headline is measured with)
mkdir -p gpurun_out
FLAGS="--steps 8 --warmup 2 --cpu-seconds 0 --others= --act-shape= --sharded= --no-roofline"
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout 200 python bench.py $FLAGS > gpurun_out/abenv2_$i.json 2> gpurun_out/abenv2_$i.err < /dev/null
  echo -n "[$cfg] "; python tools/bench_line.py gpurun_out/abenv2_$i.json | cut -c1-120
done
