#!/bin/bash
# Everything profiles/ is built from, in one GPU-box call (outputs under gpurun_out/, copied to profiles/ by hand):
#   1. the default bench line;
#   2. the same workload on ONE stream under rocprofv3 --kernel-trace --stats (with two units in flight the
#      kernels of the two streams overlap and every per-kernel duration is stretched by the sharing, so the
#      average the roofline is priced on is checked against the one-stream trace);
#   3. FETCH_SIZE / WRITE_SIZE counter passes (separate --pmc runs) + their digest.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=${PROFILE_BATCH:-32}
if [ -z "$SKIP_BENCH" ]; then
python bench.py > gpurun_out/r01_bench_default.json 2> gpurun_out/r01_bench_default.err
fi
rm -rf gpurun_out/prof_r01
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01 -o r01 -- python bench.py --streams 1 --steps 6 --warmup 2 --cpu-seconds 0 > gpurun_out/r01_bench_under_rocprof.json 2> gpurun_out/r01_bench_under_rocprof.err
bash tools/pmc_level.sh --batch $B
python tools/bench_line.py gpurun_out/r01_bench_default.json gpurun_out/r01_bench_under_rocprof.json
head -6 gpurun_out/prof_r01/*kernel_stats.csv | cut -c1-200
tail -3 gpurun_out/pmc_FETCH_SIZE.log | cut -c1-300
ls gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE 2>&1 | head
