#!/bin/bash
# Round-3 baseline on one box: GPU tests, bench.py, per-kernel stats of bench.py, clock reconciliation of the four-wave GEMM.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/r3_gpu_tests1.log 2>&1 || { tail -20 gpurun_out/r3_gpu_tests1.log; exit 1; }
tail -2 gpurun_out/r3_gpu_tests1.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench0.json 2> gpurun_out/r3_bench0.err || { tail -5 gpurun_out/r3_bench0.err; exit 1; }
cat gpurun_out/r3_bench0.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_prof0 -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r3_prof0.log 2>&1 || { tail -5 gpurun_out/r3_prof0.log; exit 1; }
timeout -k 10 200 python tools/quad_prof.py 256 64 8 > gpurun_out/r3_quad_clock.log 2>&1 || { tail -5 gpurun_out/r3_quad_clock.log; exit 1; }
cat gpurun_out/r3_quad_clock.log
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/r3_quad_clock_pmc -- python tools/quad_prof.py 256 > gpurun_out/r3_quad_clock_pmc.log 2>&1 || { tail -5 gpurun_out/r3_quad_clock_pmc.log; exit 1; }
cat gpurun_out/r3_quad_clock_pmc.log
ls -R gpurun_out/r3_prof0 gpurun_out/r3_quad_clock_pmc | head -30
