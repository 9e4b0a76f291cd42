#!/bin/bash
# A/B runs of bench.py under different values of one environment knob: bash tools/ab_env.sh VAR v1 v2 ...
var=$1; shift
for v in "$@"; do
  echo "== $var=$v"
  env $var=$v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --in-flight 1 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['stage_ms_per_step'])"
done
