#!/bin/bash
# Where the waves of the kernels spend their cycles (SQ wait / active counters), one small pass under timeout.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_sq3
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_sq3 -o sq -- python bench.py --streams 1 --steps 1 --warmup 1 --cpu-seconds 0 --no-roofline "$@" > gpurun_out/pmc_sq3.log 2>&1
echo "rc=$?"; tail -1 gpurun_out/pmc_sq3.log | cut -c1-200
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_sq3/*counter_collection.csv')
if not f:
    raise SystemExit('no counter file')
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    if 'le_level' not in r['Kernel_Name'] and 'bc_step' not in r['Kernel_Name']:
        continue
    k = (r['Kernel_Name'].split('(')[0][-18:], r['Grid_Size'])
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVES':
        n[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:10]:
    w = max(c.get('SQ_WAVES', 0), 1)
    print('%-18s grid %-9s disp %4d | wave-cycles/wave %7.0f insts/wave %6.0f ifetch/wave %5.0f | ifetch latency %.0f | vmem insts/wave %4.0f vmem latency %.0f' % (
        k[0], k[1], n[k], c.get('SQ_WAVE_CYCLES', 0) / w, c.get('SQ_INSTS', 0) / w, c.get('SQ_IFETCH', 0) / w,
        c.get('SQ_IFETCH_LEVEL', 0) / max(c.get('SQ_IFETCH', 1), 1), c.get('SQ_INSTS_VMEM', 0) / w,
        c.get('SQ_INST_LEVEL_VMEM', 0) / max(c.get('SQ_INSTS_VMEM', 1), 1)))
PY
