#!/bin/bash
# Same-box alternating A/B of two builds of the library (IVFHNSW_HIP_LIB picks the file): walk time moves by ~10 %
# from box to box (clocks), so only runs that alternate on ONE box compare.
#   bash tools/ab_lib.sh build_ab/old.so build_ab/new.so [rounds] [bench args...]
cd "$(dirname "$0")/.."
A=$1; B=$2; R=${3:-3}; shift 3
for r in $(seq 1 $R); do
  for L in "$A" "$B"; do
    echo "== $L (round $r)"
    IVFHNSW_HIP_LIB=$PWD/$L python bench.py --no-cpu-baseline --no-secondary --in-flight 1 "$@" 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['stage_ms_per_step'], j.get('parity'))"
  done
done
