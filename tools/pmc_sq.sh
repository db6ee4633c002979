#!/bin/bash
# Instruction mix of the kernels (SQ counters), one pass, small run, under timeout.  usage: tools/pmc_sq.sh [bench flags]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_sq
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_sq -o sq -- python bench.py --streams 1 --steps 1 --warmup 1 --cpu-seconds 0 --no-roofline "$@" > gpurun_out/pmc_sq.log 2>&1
echo "rc=$?"; tail -2 gpurun_out/pmc_sq.log | cut -c1-200
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_sq/*counter_collection.csv')
if not f:
    raise SystemExit('no counter file')
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = (r['Kernel_Name'].split('(')[0], r['Grid_Size'])
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVES':
        n[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU', 0))[:14]:
    w = max(c.get('SQ_WAVES', 0), 1)
    print('%-28s grid %-9s disp %4d waves/disp %7.0f | per wave: VALU %6.0f SALU %6.0f LDS %5.0f SMEM %5.0f | wait_any/busy %.2f valu_active/busy %.2f' % (
        k[0][-28:], k[1], n[k], w / max(n[k], 1), c.get('SQ_INSTS_VALU', 0) / w, c.get('SQ_INSTS_SALU', 0) / w, c.get('SQ_INSTS_LDS', 0) / w,
        c.get('SQ_INSTS_SMEM', 0) / w, c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_BUSY_CYCLES', 1), 1), c.get('SQ_ACTIVE_INST_VALU', 0) / max(c.get('SQ_BUSY_CYCLES', 1), 1)))
PY
