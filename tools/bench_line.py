#!/usr/bin/env python3
"""Print a one-line digest of bench.py JSON files (tuning aid)."""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().splitlines()[0])
        r = d.get('roofline') or {}
        alone = d['config'].get('one_unit_alone_ms', {})
        print('%s value %.3e ms/step %.3f lat %.3f le %.3f bc %.3f | level us %.2f GB/s %.0f frac %.3f sweep_us %.1f ctl %.1f | %s' % (
            f.split('/')[-1], d['value'], d['ms_per_step'], d['config']['single_pass_latency_ms'], alone.get('equalization', 0), alone.get('bias_correction', 0),
            r.get('us_per_launch', 0), r.get('achieved', 0), r.get('frac', 0), r.get('sweep_wall_us', 0),
            r.get('control_us_per_sweep', 0),
            ' '.join('%.1f' % l['us'] for l in r.get('levels', []))))
        ar = d['config'].get('activation_ranges')
        if ar:
            print('    act: ' + ' '.join('%s %.1f us' % (k['kernel'] if isinstance(k, dict) and 'kernel' in k else str(i), k.get('us', 0)) for i, k in enumerate(ar if isinstance(ar, list) else ar.get('kernels', []))))
    except Exception as e:
        print(f, 'failed', e)
