#!/bin/bash
# Three counter passes over a short bench.py run (each a run of its own, counters only + kernel trace for the durations).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for pass in "util:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  tag=${pass%%:*}; ctrs=${pass#*:}
  rm -rf gpurun_out/r3_pmc_$tag
  timeout -s KILL 280 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d gpurun_out/r3_pmc_$tag -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3_pmc_$tag.log 2>&1 || { tail -5 gpurun_out/r3_pmc_$tag.log; exit 1; }
  find gpurun_out/r3_pmc_$tag -name "*counter_collection.csv" -exec cp {} gpurun_out/r3_pmc_$tag.csv \;
  find gpurun_out/r3_pmc_$tag -name "*.csv" -not -name "*counter_collection.csv" -delete
  ls -la gpurun_out/r3_pmc_$tag.csv
done
python tools/pmc_summary.py gpurun_out/r3_pmc_util.csv gpurun_out/r3_pmc_fetch.csv gpurun_out/r3_pmc_write.csv gpurun_out/r03
rm -rf gpurun_out/r3_pmc_util gpurun_out/r3_pmc_fetch gpurun_out/r3_pmc_write gpurun_out/r3_pmc_util.csv gpurun_out/r3_pmc_fetch.csv gpurun_out/r3_pmc_write.csv
