#!/bin/bash
# A/B of environment switches of the library on ONE GPU box, alternating rounds (boxes differ by a few per cent: only numbers of
# one call compare).  usage: tools/ab_env.sh "VAR=val [VAR=val]" ["VAR=val" ...]   (an empty string = the defaults)
#   e.g.  gpurun -- 'bash tools/ab_env.sh "" "DFQ_LE_CF=0" "DFQ_LE_CF_GROUP=4"'        (round 6: the free-running segments)
#         DFQ_HIP_LIB=$PWD/variants/libdfq_hip_x.so in a setting selects a variant build (make BUILD=build_x OUT=... EXTRA=-D...)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
FLAGS="${AB_FLAGS:---steps 8 --warmup 3 --cpu-seconds 0 --others= --act-shape= --sharded= --distill= --pcie= --lazy-steps 0}"
for round in 1 2; do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    env $cfg timeout 300 python bench.py $FLAGS > gpurun_out/abenv_${i}_$round.json 2> gpurun_out/abenv_${i}_$round.err < /dev/null
    echo -n "round $round [$cfg] "; python tools/bench_line.py gpurun_out/abenv_${i}_$round.json | head -1 | cut -c1-220
  done
done
