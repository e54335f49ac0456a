#!/bin/bash
# A/B two environment settings of bench.py in ONE gpurun call (same device), interleaved rounds.
# usage: tools/ab_bench.sh "ENV_A=1" "ENV_B=1" [rounds] [extra bench args]
A="$1"; B="$2"; R="${3:-3}"; shift 3 || true
for i in $(seq 1 $R); do
  for cfg in "$A" "$B"; do
    v=$(env $cfg python bench.py --steps 300 --warmup 30 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "round $i [$cfg] $v"
  done
done
