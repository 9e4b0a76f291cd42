#!/bin/bash
# A/B runs of bench.py, one "VAR=val VAR2=val2" setting per argument
for v in "$@"; do
  echo "== $v"
  env $v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --in-flight 1 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['stage_ms_per_step'])"
done
