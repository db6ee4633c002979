#!/bin/bash
# tuning aid: per-level kernel time of the LE sweep with parts of the tile kernel switched off
for ab in 0 2 4 8 6 14 30 31; do
  echo "== ablate $ab"
  DFQ_LE_ABLATE=$ab python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --sweeps 47 --force-sweeps 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d['roofline']
print('ms/step %.3f  level us: %s  control %.1f' % (d['ms_per_step'], ' '.join('%.1f' % l['us'] for l in r['levels']), r['control_us_per_sweep']))"
done
