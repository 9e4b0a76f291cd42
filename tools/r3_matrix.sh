#!/bin/bash
# round 3: the step's stages on {iid, clustered} tables x {plain k-NN graph, insertion-loop graph}, IVFADC and Grouping
cd "$(dirname "$0")/.."
O=gpurun_out/r3_matrix; mkdir -p $O
B="python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary --in-flight 1 --sustain-s 0"
run() { # name, env..., workload
  n=$1; shift; w=$1; shift
  env "$@" timeout -k 10 400 $B --workload $w > $O/$n.json 2> $O/$n.err || echo "FAILED $n"
  python - $O/$n.json $n <<'PY'
import json, sys
try:
    o = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    sp = o.get("split_batch") or {}
    print("%-28s value %.2f M  stages %s  scan frac %.3f  split %s" % (sys.argv[2], o["value"] / 1e6, o["stage_ms_per_step"], o["roofline"]["frac"], sp.get("queries_per_s")), flush=True)
except Exception as e:
    print(sys.argv[2], "no line:", e)
PY
}
run iid_knn synthetic-1B-pq16-nc993127-nprobe32 IVFHNSW_BENCH_GRAPH=knn
run iid_insert synthetic-1B-pq16-nc993127-nprobe32 IVFHNSW_BENCH_GRAPH=insert
run clu_insert clustered-1B-pq16-nc993127-nprobe32 IVFHNSW_BENCH_GRAPH=insert
run grp_iid_insert grouping-1B-pq16-nc993127-nsubc64-opq-pruning IVFHNSW_BENCH_GRAPH=insert
run grp_clu_insert clustered-grouping-1B-pq16-nc993127-nsubc64-opq-pruning IVFHNSW_BENCH_GRAPH=insert
