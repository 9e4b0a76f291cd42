#!/bin/bash
# N ranks of bench.py on ONE GPU over gloo against the single-rank run of the same corpus and batch: labels and
# distance bits must agree (N <= 5: the box allows six processes on its GPU and the launcher counts).
# usage: bash tools/rehearse_ranks.sh <N> [workload] [extra bench args, e.g. --scaling strong, --list-shards 2]
# (with --list-shards S the comparison is rank 0's replica group: S shards, its own batch)
set -e
cd "$(dirname "$0")/.."
N=${1:-4}; W=${2:-synthetic-100M-pq16-nc131072-nprobe32}; shift; shift || true
B=$(python - "$W" "$N" "$@" <<'PY'
import sys; sys.path.insert(0, '.')
import bench
w, n, rest = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
s = int(rest[rest.index("--list-shards") + 1]) if "--list-shards" in rest else n  # rank 0's group: S shards, its own batch
print(bench.STRONG_BATCH // (n // s) if "strong" in rest else bench.WORKLOADS[w][7] * s)
PY
)
mkdir -p gpurun_out
python bench.py --gpus 1 --batch $B --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --sustain-s 0 --in-flight 1 --workload $W --dump gpurun_out/reh_1.npz > gpurun_out/reh_1.log 2>&1
echo "[rehearse] single rank done (batch $B)"
IVFHNSW_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus $N --steps 3 --warmup 1 --sustain-s 0 --workload $W --dump gpurun_out/reh_n.npz "$@" > gpurun_out/reh_n.log 2>&1
echo "[rehearse] $N ranks done"; tail -1 gpurun_out/reh_n.log | cut -c1-600
python - <<'PY'
import numpy as np
a = np.load('gpurun_out/reh_1.npz'); b = np.load('gpurun_out/reh_n.npz')
same_l = (a['labels'] == b['labels']).all(); same_d = (a['dist'].view('u4') == b['dist'].view('u4')).all()
print("[rehearse] %d queries: labels equal %s, distance bits equal %s" % (len(a['labels']), same_l, same_d))
raise SystemExit(0 if same_l and same_d else 1)
PY
rm -f gpurun_out/reh_1.npz gpurun_out/reh_n.npz
