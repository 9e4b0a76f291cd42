#!/bin/bash
# A/B of the fused table+scan kernel against lut_kernel + scan_k1_kernel on the GPU box (gpurun -- 'bash tools/ab_fused.sh')
cd "$(dirname "$0")/.."
O=gpurun_out/abf; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-secondary --in-flight 1 --sustain-s 0.5"
for w in ${AB_WORKLOADS:-synthetic-1B-pq16-nc993127-nprobe32 grouping-1B-pq16-nc993127-nsubc64-opq-pruning}; do
  for f in ${AB_FORMS:-1 0}; do
    IVFHNSW_SCAN_FUSED=$f timeout -k 10 300 $B --workload $w > $O/$w.f$f.json 2> $O/$w.f$f.err || echo "FAILED $w f$f"
    python3 - $O/$w.f$f.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], "q/s %.0f" % j["value"], "ms %.4f" % j["ms_per_step"], j["roofline"]["kernel"], "frac", j["roofline"]["frac"], j["stage_ms_per_step"])
PY
  done
done
