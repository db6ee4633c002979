#!/bin/bash
# round 5: the bench's batch of 32 and its single-network legs under a list of environment settings ("" = defaults first), alternating,
# ROUNDS (default 2) rounds.   usage: tools/gpu_r05_bench_env_ab.sh "ENV1=a" "ENV1=b ENV2=c" ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
F="--cpu-seconds 0 --others=resnet18,deeplab_mnv2:60 --act-shape= --sharded= --lazy-steps 0 --pcie= --distill= --steps 8 --warmup 2"
{
for round in $(seq 1 ${ROUNDS:-2}); do
  for s in "" "$@"; do
    echo "== [$s] (round $round)"
    env $s timeout 300 python bench.py $F 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); L=d['latency']; u=d['config'].get('one_unit_alone_ms', {})
        print('value %.4g  batch le %.3f bc %.4f ms | single pass %.4f = le %.4f + bc %.4f | others %s' % (d['value'], u.get('equalization', 0), u.get('bias_correction', 0), L['single_network_pass_ms'], L['equalization_gpu_ms'], L['bias_correction_gpu_ms'], [(o['net'], round(o['ms'],4), round(o.get('bias_correction_ms',0),4)) for o in d['config']['others']]))
"
  done
done
} > gpurun_out/r05/bench_env_ab.txt 2>&1
cat gpurun_out/r05/bench_env_ab.txt
