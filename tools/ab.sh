#!/bin/bash
# A/B runs of bench.py on the GPU box, one environment setting per argument (knobs are read once per process):
#   bash tools/ab.sh [-w <workload>] "X=1" "IVFHNSW_WALK_MERGE=0" ...
# Prints queries/s and the per-stage milliseconds of each run.
cd "$(dirname "$0")/.."
W=""
if [ "$1" = "-w" ]; then W="--workload $2"; shift 2; fi
for v in "$@"; do
  echo "== $v"
  env $v python bench.py --no-cpu-baseline --in-flight 1 $W 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['stage_ms_per_step'], 'scan frac', j['roofline']['frac'], 'launch ms', j['roofline']['avg_launch_ms'])"
done
