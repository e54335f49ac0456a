#!/bin/bash
# Interleaved A/B/... of bench.py under several environment settings in ONE gpurun call.
# usage: tools/ab_multi.sh ROUNDS "ENV_A=1" "ENV_B=1 ENV_C=2" ...   (use "X=0" for the plain default)
R="$1"; shift
for i in $(seq 1 $R); do
  for cfg in "$@"; do
    v=$(env $cfg python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "round $i [$cfg] $v"
  done
done
