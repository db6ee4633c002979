#!/bin/bash
# HBM traffic of le_level_kernel from the TCC counters (separate --pmc passes, MI355X_MICROARCH.md "HBM"):
# FETCH_SIZE counts 64-B units for 128-B requests on gfx950 -> doubled for wide coalesced reads.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o pmc -- python bench.py --streams 1 --steps 4 --warmup 2 --cpu-seconds 0 --no-roofline > gpurun_out/pmc_$c.log 2>&1
  ls gpurun_out/pmc_$c | head
done
python - <<'PY'
import csv, glob, collections
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob('gpurun_out/pmc_%s/*counter_collection.csv' % c)
    if not files:
        print(c, 'no counter file'); continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(files[0])):
        if r.get('Counter_Name') == c:
            k = r['Kernel_Name'].split('(')[0]
            acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
    for k, (v, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:6]:
        print(c, k, 'dispatches', n, 'mean per dispatch', v / n)
PY
