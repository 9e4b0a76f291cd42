#!/bin/bash
# A/B runs of bench.py on a named workload: bash tools/ab_env3.sh <workload> "VAR=val ..." ...
w=$1; shift
for v in "$@"; do
  echo "== $v"
  env $v python bench.py --no-cpu-baseline --in-flight 1 --workload $w 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['stage_ms_per_step'])"
done
