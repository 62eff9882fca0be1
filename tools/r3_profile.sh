#!/bin/bash
# bench.py, then its per-kernel stats under rocprofv3 (kernel trace only), on one box.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err || { tail -5 gpurun_out/r3_bench1.err; exit 1; }
cat gpurun_out/r3_bench1.json
rm -rf gpurun_out/r3_prof1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_prof1 -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r3_prof1.log 2>&1 || { tail -5 gpurun_out/r3_prof1.log; exit 1; }
find gpurun_out/r3_prof1 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3_bench_kernel_stats.csv \;
find gpurun_out/r3_prof1 -name "*kernel_trace.csv" -delete
tail -1 gpurun_out/r3_prof1.log
