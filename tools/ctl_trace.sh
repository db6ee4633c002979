cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ctl_trace
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ctl_trace -o t -- python tools/pmc_unit.py --batch 64 --sweeps 24 --le-only > gpurun_out/ctl_trace.log 2>&1
T=$(find gpurun_out/ctl_trace -name "*kernel_trace.csv" | head -1)
python - "$T" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = max(i for i, r in enumerate(rows) if 'le_prepare_kernel' in r['Kernel_Name'])
t0 = int(rows[idx]['Start_Timestamp']); prev=t0
for r in rows[idx:idx+70]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('{:9.1f} dur {:7.1f} gap {:6.1f} {}'.format((s-t0)/1e3, (e-s)/1e3, (s-prev)/1e3, r['Kernel_Name'][:40]))
    prev=e
PY
rm -rf gpurun_out/ctl_trace
