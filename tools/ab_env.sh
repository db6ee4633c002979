#!/bin/bash
# A/B of plan-time environment switches of the library on ONE GPU box.
# usage: tools/ab_env.sh "VAR=val [VAR=val]" ["VAR=val" ...]   (an empty string = defaults)
mkdir -p gpurun_out
FLAGS="--steps 4 --warmup 1 --cpu-seconds 0 --others= --act-shape= --sharded= --streams 1"
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout 200 python bench.py $FLAGS > gpurun_out/abenv_$i.json 2> gpurun_out/abenv_$i.err < /dev/null
  echo -n "[$cfg] "; python tools/bench_line.py gpurun_out/abenv_$i.json | cut -c1-200
done
