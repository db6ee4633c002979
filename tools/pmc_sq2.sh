#!/bin/bash
# Where the waves of the kernels spend their cycles (SQ wait / active counters), one small pass under timeout.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_sq2
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VMEM_WR_TA_DATA_FIFO_FULL --kernel-trace --output-format csv -d gpurun_out/pmc_sq2 -o sq -- python bench.py --streams 1 --steps 1 --warmup 1 --cpu-seconds 0 --no-roofline "$@" > gpurun_out/pmc_sq2.log 2>&1
echo "rc=$?"; tail -1 gpurun_out/pmc_sq2.log | cut -c1-200
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmc_sq2/*counter_collection.csv')
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
    wc = max(c.get('SQ_WAVE_CYCLES', 1), 1)
    print('%-18s grid %-9s disp %4d | wave-cycles/wave %7.0f | of which wait_any %.2f wait_inst_any %.2f vmem %.3f lds %.3f valu %.3f | wr_fifo_full/wave %.0f' % (
        k[0], k[1], n[k], wc / w, c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get('SQ_ACTIVE_INST_VMEM', 0) / wc,
        c.get('SQ_ACTIVE_INST_LDS', 0) / wc, c.get('SQ_ACTIVE_INST_VALU', 0) / wc, c.get('SQ_VMEM_WR_TA_DATA_FIFO_FULL', 0) / w))
PY
