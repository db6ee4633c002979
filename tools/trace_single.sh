#!/bin/bash
# Kernel timeline of ONE network's pass (tools/lat.py under rocprofv3 --kernel-trace): start / duration / gap of the last pass's launches.
# usage: tools/trace_single.sh resnet18   (GPU box)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
NET=${1:-resnet18}
rm -rf gpurun_out/trace_single
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_single -o t -- python tools/lat.py $NET > gpurun_out/trace_single.log 2>&1
T=$(find gpurun_out/trace_single -name "*kernel_trace.csv" | head -1)
python - "$T" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last pass: walk back from the end to the last le_prepare / res_prepare launch
idx = max(i for i, r in enumerate(rows) if 'prepare_kernel' in r['Kernel_Name'])
t0 = int(rows[idx]['Start_Timestamp'])
prev_end = t0
for r in rows[idx:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('{:9.2f} us  dur {:7.2f}  gap {:6.2f}  {}'.format((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r['Kernel_Name'][:90]))
    prev_end = e
PY
rm -rf gpurun_out/trace_single
