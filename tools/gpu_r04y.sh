#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_fl
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fl -o f -- python bench.py --streams 1 --steps 4 --warmup 2 --cpu-seconds 0 --sharded "" --distill "" --pcie "" --others "" --lazy-steps 0 --no-roofline --act-shape "" > /dev/null 2>&1
S=$(find gpurun_out/prof_fl -name "*kernel_stats.csv" | head -1)
python - "$S" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0][-40:]
    if any(k in n for k in ('le_', 'bc_')):
        print('%-42s calls %5s avg %8.1f min %8.1f max %8.1f us' % (n, r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
rm -rf gpurun_out/prof_fl
