#!/bin/bash
# Same-box comparison of bench.py under several settings of one environment variable: bench_env_ab.sh VAR v1 v2 ...
set -o pipefail
var=$1; shift
for round in 1 2; do
  for v in "$@"; do
    env $var=$v timeout -k 10 300 python -u bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', d['ms_per_step'], d['value'], d['roofline']['frac'])" || exit 1
  done
done
