#!/bin/bash
# round 5: bias correction with / without the gate in front of the slot polls: one network (tools/lat.py) and the bench's batch of 32
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
F="--cpu-seconds 0 --others=resnet18,deeplab_mnv2:60 --act-shape= --sharded= --lazy-steps 0 --pcie= --distill= --steps 8 --warmup 2"
{
for round in 1 2; do
  for s in "DFQ_BC_GATE=1" "DFQ_BC_GATE=0" "$@"; do
    echo "== [$s] (round $round)"
    env $s timeout 120 python tools/lat.py mobilenet_v2 deeplab_mnv2:60 resnet18 2>/dev/null
    env $s timeout 200 python bench.py $F 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); L=d['latency']
        print('bench: value %.4g  bc_batch_ms %s  single pass %.4f = le %.4f + bc %.4f | others %s' % (d['value'], d['config'].get('one_unit_alone_ms',{}).get('bias_correction'), L['single_network_pass_ms'], L['equalization_gpu_ms'], L['bias_correction_gpu_ms'], [(o['net'], round(o['ms'],4), round(o.get('bias_correction_ms',0),4)) for o in d['config']['others']]))
"
  done
done
} > gpurun_out/r05/bc_ab.txt 2>&1
cat gpurun_out/r05/bc_ab.txt
