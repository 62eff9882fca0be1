#!/bin/bash
# Predictor tests, the permutation check, the rollout bench, and its kernel-trace profile.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_predictor_gpu.py tests/test_configs_gpu.py::test_config5_full_size_rollout_properties -x -q --timeout 300 --timeout-method thread > gpurun_out/t1.log 2>&1 || { tail -30 gpurun_out/t1.log; exit 1; }
tail -3 gpurun_out/t1.log
timeout -k 10 200 python -u tools/rollout_perm_diff.py 8 2 4 > gpurun_out/perm.log 2>&1 || { tail -30 gpurun_out/perm.log; exit 1; }
cat gpurun_out/perm.log
timeout -k 10 200 python -u tools/rollout_bench.py 64 4 > gpurun_out/rollout.log 2>&1 || { tail -30 gpurun_out/rollout.log; exit 1; }
cat gpurun_out/rollout.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rollout_prof -o rollout -- python3 $GRAFT_REPO_ROOT/tools/rollout_bench.py 64 4 > $GRAFT_REPO_ROOT/gpurun_out/rollout_prof.log 2>&1 || { tail -30 $GRAFT_REPO_ROOT/gpurun_out/rollout_prof.log; exit 1; }
cd $GRAFT_REPO_ROOT && find gpurun_out/rollout_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/rollout_kernel_stats.csv \; && find gpurun_out/rollout_prof -name "*kernel_trace.csv" -delete
