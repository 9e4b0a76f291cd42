#!/bin/bash
# One-off counter probes on the GPU box: bash tools/pmc_probe.sh "<counters>" [more bench args]
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/probe
rm -rf "$O"; mkdir -p "$O"
B="python3 $PWD/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-split --in-flight 1 $BENCH_ARGS"
cd /tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $set -d "$O/p$i" -o run -- $B > /dev/null 2> "$O/p$i.err" || echo "pass $i failed"
done
rocprofv3 -L > "$O/avail.txt" 2>&1 || true
find "$O" -type f ! -name "*counter_collection.csv" ! -name "*.err" ! -name "avail.txt" -delete
python3 - "$O" <<'PY'
import csv,glob,sys,collections
for f in sorted(glob.glob(sys.argv[1]+'/p*/**/*counter_collection.csv',recursive=True)):
    acc=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'].split('(')[0][:40],r['Counter_Name'])
        acc[k][0]+=1; acc[k][1]+=float(r['Counter_Value'])
    for k,v in sorted(acc.items()):
        if 'walk' in k[0] or 'scan' in k[0] or 'plan' in k[0]:
            print(k[0],k[1],v[0],v[1]/v[0])
PY
