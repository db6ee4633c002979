#!/bin/bash
# the headline alone, twice (tuning aid: quick check of a change to the sweep kernel)
mkdir -p gpurun_out/quick
F="--cpu-seconds 0 --others= --act-shape= --sharded= --lazy-steps 0 --pcie= --distill="
for r in 1 2; do
  timeout 200 python bench.py $F > gpurun_out/quick/q$r.json 2>/dev/null < /dev/null
  python tools/bench_line.py gpurun_out/quick/q$r.json | head -1 | cut -c1-200
done
