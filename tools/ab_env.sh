#!/bin/bash
# A/B of one environment knob on the bench (gpurun -- 'bash tools/ab_env.sh VAR "v1 v2" [workloads...]')
cd "$(dirname "$0")/.."
VAR=$1; VALS=$2; shift; shift
O=gpurun_out/abenv; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-secondary --in-flight 1 --sustain-s 0.5"
for w in ${@:-synthetic-1B-pq16-nc993127-nprobe32}; do
  for v in $VALS; do
    env $VAR=$v timeout -k 10 300 $B --workload $w > $O/$w.$VAR$v.json 2> $O/$w.$VAR$v.err || echo "FAILED $w $VAR=$v"
    python3 - $O/$w.$VAR$v.json $VAR=$v <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
print(sys.argv[2], j["config"]["workload"], "q/s %.0f" % j["value"], "ms %.4f" % j["ms_per_step"], "sustained %.0f" % j["sustained"]["queries_per_s"], j["stage_ms_per_step"])
PY
  done
done
