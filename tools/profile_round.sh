#!/bin/bash
# Everything profiles/ is built from, in one GPU-box call (outputs under gpurun_out/, copied to profiles/ by hand):
#   1. the default bench line;
#   2. the same workload on ONE stream under rocprofv3 --kernel-trace --stats (with two units in flight the
#      kernels of the two streams overlap and every per-kernel duration is stretched by the sharing, so the
#      average the roofline is priced on is checked against the one-stream trace); the run also executes the
#      single-network latency probe (le_resident_kernel), the other configs and the activation-range kernels;
#   3. FETCH_SIZE / WRITE_SIZE counter passes (separate --pmc runs) at the bench's own batch size + their digest.
# Every step runs under `timeout`; nothing here reads stdin.
R=${ROUND:-r06}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=${PROFILE_BATCH:-64}
if [ -z "$SKIP_BENCH" ]; then
timeout 600 python bench.py > gpurun_out/${R}_bench_default.json 2> gpurun_out/${R}_bench_default.err < /dev/null
fi
rm -rf gpurun_out/prof_$R
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$R -o $R -- python bench.py --streams 1 --steps 6 --warmup 2 --cpu-seconds 0 --sharded "" --distill "" --pcie "" > gpurun_out/${R}_bench_under_rocprof.json 2> gpurun_out/${R}_bench_under_rocprof.err < /dev/null
echo "rocprof rc=$?"
S=$(find gpurun_out/prof_$R -name "*kernel_stats.csv" | head -1)
T=$(find gpurun_out/prof_$R -name "*kernel_trace.csv" | head -1)
D=$(find gpurun_out/prof_$R -name "*domain_stats.csv" | head -1)
[ -n "$S" ] && cp "$S" gpurun_out/${R}_bench_kernel_stats.csv && head -12 "$S" | cut -c1-180
[ -n "$D" ] && cp "$D" gpurun_out/${R}_bench_domain_stats.csv
[ -n "$T" ] && timeout 120 python tools/rocprof_digest.py "$T" gpurun_out/${R}_bench_under_rocprof.json > gpurun_out/${R}_level_kernel_by_launch.csv < /dev/null
[ -n "$T" ] && rm -f "$T"
PMC_TIMEOUT=200 bash tools/pmc_level.sh --batch $B --sweeps 8 --no-lazy < /dev/null     # (8: one whole group of the free-running layers; --no-lazy: with the opt-in lazy-scale engine in the process a counter pass at batch 64 aborts inside rocprofv3 -- "AQL packet is malformed")
timeout 120 python tools/pmc_digest.py gpurun_out gpurun_out/${R}_bench_under_rocprof.json gpurun_out/${R}_pmc_summary.json < /dev/null | tail -12
timeout 60 python tools/bench_line.py gpurun_out/${R}_bench_default.json gpurun_out/${R}_bench_under_rocprof.json < /dev/null
tail -3 gpurun_out/pmc_FETCH_SIZE.log | cut -c1-300
