#!/bin/bash
# HBM traffic of le_level_kernel from the TCC counters (separate --pmc passes, MI355X_MICROARCH.md "HBM").
# usage: tools/pmc_level.sh [--batch B] [--sweeps S]   (run on the GPU box; digest with tools/pmc_digest.py)
# The profiled process is tools/pmc_unit.py: one batched unit, a few forced sweeps (bench.py's set-up is far too many dispatches
# for a counter pass).
# Every pass runs under `timeout`: a counter pass that aborts can leave rocprofv3 waiting forever.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout ${PMC_TIMEOUT:-150} rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -o pmc -- python tools/pmc_unit.py "$@" > gpurun_out/pmc_$c.log 2>&1 < /dev/null
  echo "pmc $c rc=$?"
done
