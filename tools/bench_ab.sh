#!/bin/bash
# Same-box A/B of the whole training step: bench.py with the library given as $1 against the in-tree build, twice each.
set -o pipefail
mkdir -p gpurun_out
for i in 1 2; do
  VIT_AMD_LIB=$1 timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('base', d['ms_per_step'], d['value'])" || exit 1
  timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('new ', d['ms_per_step'], d['value'], d['roofline']['frac'])" || exit 1
done
