#!/bin/bash
# FETCH_SIZE calibration for the walk's access shapes (tools/fetch_probe.hip): the program itself after `--` (no wrapper)
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
O=$ROOT/gpurun_out/fetch_probe
rm -rf "$O"; mkdir -p "$O"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$O/p" -o run -- $ROOT/tools/fetch_probe.bin 8 > "$O/stdout.txt" 2> "$O/err.txt"
cat "$O/stdout.txt"
python3 - "$O" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/p/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE' and 'probe_' in r['Kernel_Name']:
            print(r['Kernel_Name'].split('(')[0], 'FETCH_SIZE', r['Counter_Value'], 'KB =', float(r['Counter_Value']) * 1024 / 1e9, 'GB')
PY
find "$O" -name "*.csv" -size +1M -delete
