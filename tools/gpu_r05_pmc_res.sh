#!/bin/bash
# round 5: SQ counters of the resident kernel on ONE MobileNetV2 (47 pinned sweeps): where do its waves' cycles go -- instruction fetch?
# LDS?  waiting?  Separate --pmc passes (never with a trace domain), each under timeout.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05/pmc_res
rocprofv3 -L > gpurun_out/r05/pmc_res/counters.txt 2>&1
grep -o "SQC\?_[A-Z0-9_]*" gpurun_out/r05/pmc_res/counters.txt | sort -u | tr '\n' ' ' | cut -c1-6000
echo
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES" \
           "SQ_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM" \
           "SQ_WAVES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
           "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d gpurun_out/r05/pmc_res/p$i -o sq -- python tools/lat.py mobilenet_v2:47 > gpurun_out/r05/pmc_res/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(collections.Counter)
for f in glob.glob('gpurun_out/r05/pmc_res/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-40:]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[k][r['Counter_Name']] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:6]:
    d = max(n[k].get('SQ_WAVE_CYCLES', 1), 1)
    print('==', k, 'dispatches', d)
    for name in sorted(c):
        print('   %-32s %16.0f   per dispatch %14.1f' % (name, c[name], c[name] / max(n[k][name], 1)))
PY
