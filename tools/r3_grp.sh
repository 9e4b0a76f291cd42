#!/bin/bash
# Grouping plan: direct vs DEDUPE on iid and clustered tables (stage times of the one-part shape)
cd "$(dirname "$0")/.."
for w in grouping-1B-pq16-nc993127-nsubc64-opq-pruning clustered-grouping-1B-pq16-nc993127-nsubc64-opq-pruning; do
  for dd in 0 1 auto; do
    e="X=1"; [ $dd != auto ] && e="IVFHNSW_PLAN_DEDUPE=$dd"
    env $e python bench.py --no-cpu-baseline --no-secondary --in-flight 1 --sustain-s 0 --steps 100 --no-split --workload $w 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w dedupe=$dd', j['value'], j['stage_ms_per_step'])"
  done
done
