#!/bin/bash
# usage: tools/gpu_prof.sh tag -- command...   (rocprofv3 kernel stats of a command, digest printed)
tag=$1; shift; shift
mkdir -p gpurun_out/$tag
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp PYTHONPATH=$R
cmd="$*"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -o prof -- bash -c "cd $R && $cmd" > $R/gpurun_out/$tag/run.log 2>&1 )
cd $R
tail -4 gpurun_out/$tag/run.log
f=$(find gpurun_out/$tag -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print('{:60s} calls {:>6s} total_us {:>10.1f} avg_us {:>9.2f} {:>6s}%'.format(r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3, r['Percentage']))
PY
t=$(find gpurun_out/$tag -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python - "$t" <<'PY'
import csv, sys, collections
# per (kernel, grid size): calls and mean duration -- separates the launches of one kernel by what they cover
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    key = (r['Kernel_Name'][:40], r.get('Grid_Size_X', r.get('Grid_Size', '?')))
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    a = agg.setdefault(key, [0, 0]); a[0] += 1; a[1] += d
for (k, g), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print('{:42s} grid {:>9s} calls {:>6d} avg_us {:>9.2f} total_ms {:>8.2f}'.format(k, g, n, tot / n / 1e3, tot / 1e6))
PY
