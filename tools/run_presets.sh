#!/bin/bash
# The other bench workloads (1B shapes on one GPU, Grouping): one JSON line each under gpurun_out/
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for w in ${PRESETS:-synthetic-1B-pq16-nc993127-nprobe32 synthetic-1B-pq16-nc993127-nprobe64 deep-1B-d96-opq-pq16-nc999973-nprobe128 grouping-1B-pq16-nc993127-nsubc64-opq-pruning}; do
  timeout -k 10 700 python3 bench.py --workload $w > gpurun_out/b_$w.json 2> gpurun_out/b_$w.err || exit 1
  echo "[presets] $w done"; tail -c 200 gpurun_out/b_$w.json
done
w=grouping-100M-pq16-nc131072-nsubc64-opq-pruning
timeout -k 10 500 python3 bench.py --workload $w > gpurun_out/b_$w.json 2> gpurun_out/b_$w.err || exit 1
echo "[presets] $w done"
