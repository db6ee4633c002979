#!/bin/bash
# round 5: per-kernel average durations of a short one-stream bench run under rocprofv3 (the kernels named in $KERNELS)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for s in "" "$@"; do
  rm -rf gpurun_out/r05/kstats
  echo "== [$s]"
  env $s timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05/kstats -o k -- python bench.py --streams 1 --steps 4 --warmup 1 --cpu-seconds 0 --sharded "" --distill "" --pcie "" --others "" --act-shape "" --lazy-steps 0 > /dev/null 2>&1 < /dev/null
  S=$(find gpurun_out/r05/kstats -name "*kernel_stats.csv" | head -1)
  python - "$S" <<'PY'
import csv, sys
for r in csv.reader(open(sys.argv[1])):
    if any(k in r[0] for k in ('le_control', 'le_level', 'bc_chain', 'bc_minmax', 'le_resident', 'le_bootstrap')):
        print('%-28s calls %5s avg %9.1f ns  min %8s max %8s' % (r[0].split('(')[0].replace('void ', '')[:28], r[1], float(r[3]), r[5], r[6]))
PY
done
rm -rf gpurun_out/r05/kstats
