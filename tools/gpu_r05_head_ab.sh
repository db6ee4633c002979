#!/bin/bash
# round 5: the headline alone (batch of 32, two batches in flight) under a list of environment settings, two rounds
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
F="--cpu-seconds 0 --others= --act-shape= --sharded= --lazy-steps 0 --pcie= --distill= --no-roofline"
{
for round in 1 2; do
  for s in "" "$@"; do
    echo -n "[$s] (round $round): "; env $s timeout 200 python bench.py $F 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); print('value %.4g  ms/step %.3f  streams %s' % (d['value'], d['ms_per_step'], d['config'].get('units_in_flight_per_gpu')))
"
  done
done
} > gpurun_out/r05/head_ab.txt 2>&1
cat gpurun_out/r05/head_ab.txt
