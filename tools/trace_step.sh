#!/bin/bash
# kernel timeline of a few default-path steps (rocprofv3 --kernel-trace): where the step's wall time goes between kernels
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
O=$ROOT/gpurun_out/trace
rm -rf "$O"; mkdir -p "$O"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$O/t" -o run -- python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --sustain-s 0 --in-flight 1 --no-one-part "$@" > "$O/bench.json" 2> "$O/err.txt"
python3 - "$O" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/t/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('ivfhnsw_gpu_impl::', ''), r.get('Stream_Id', '?')))
rows.sort()
# the last 3 steps: find the last 6 scan_k1 launches
scans = [i for i, r in enumerate(rows) if r[2].startswith('scan_k1_kernel')]
first = None
walks = [i for i, r in enumerate(rows) if r[2].startswith('hnsw_walk_kernel<2, 4, 10')]
i0 = walks[-4]   # first walk of the second-to-last step
t0 = rows[i0][0]
for r in rows[i0:]:
    print("%9.1f us  +%8.1f us  %-52s stream %s" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[2][:52], r[3]))
PY
