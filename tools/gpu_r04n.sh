#!/bin/bash
# round 4: resident engine, environment A/Bs on one box (launch order, cooperative launch, speculation depth)
for rep in 1 2; do
for cfg in "BASE=1" "DFQ_RES_ORDER=0" "DFQ_COOPERATIVE=0" "DFQ_RES_SPEC=3" "DFQ_RES_CKPT=4"; do
  echo "== $cfg"
  env $cfg timeout 300 python tools/lat.py mobilenet_v2 deeplab_mnv2:60 2>/dev/null
done; done
