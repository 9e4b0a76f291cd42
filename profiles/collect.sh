#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash profiles/collect.sh'): the default bench line, the same command under
# rocprofv3 --kernel-trace --stats, and four PMC passes (each in its own run, counters only with kernel trace).
# Output under gpurun_out/prof/ (scratch); condense with profiles/summarize.py and commit the summary.
set -e
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof
rm -rf "$O"; mkdir -p "$O"
timeout -k 10 500 python3 bench.py $BENCH_ARGS > "$O/bench.json" 2> "$O/bench.err"
echo "[collect] bench done"; tail -c 300 "$O/bench.json"
# D = the default path (a batch as two parts on two streams: two launches of walk, plan + tables and scan per step);
# B = the same steps in ONE part (--no-split): one launch shape per kernel, what the PMC passes are read on
D="python3 $PWD/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --sustain-s 0 --in-flight 1 --no-one-part $BENCH_ARGS"
B="$D --no-split"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o run -- $D > "$O/bench_under_rocprof.json" 2> "$O/stats.err"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats1" -o run -- $B > "$O/bench_one_part_under_rocprof.json" 2> "$O/stats1.err"
echo "[collect] stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$O/pmc_fetch" -o run -- $B --steps 2 > /dev/null 2> "$O/pmc_fetch.err"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$O/pmc_write" -o run -- $B --steps 2 > /dev/null 2> "$O/pmc_write.err"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d "$O/pmc_sq" -o run -- $B --steps 2 > /dev/null 2> "$O/pmc_sq.err"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE -d "$O/pmc_grbm" -o run -- $B --steps 2 > /dev/null 2> "$O/pmc_grbm.err"
echo "[collect] pmc done"
# keep the merge-back small: only the stats and counter tables travel
find "$O" -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "*.json" ! -name "*.err" -delete
# ... condensed on the box: the raw counter tables (torch's data-generation kernels included) exceed the merge-back limit
cd "$ROOT"
python3 profiles/summarize.py "$O"/stats/*kernel_stats.csv "$O"/pmc_fetch/*counter_collection.csv "$O"/pmc_write/*counter_collection.csv \
  "$O"/pmc_sq/*counter_collection.csv "$O"/pmc_grbm/*counter_collection.csv > "$O/summary.md"
cp "$O"/stats/*kernel_stats.csv "$O/kernel_stats.csv"
cp "$O"/stats1/*kernel_stats.csv "$O/kernel_stats_one_part.csv"
find "$O" -name "*counter_collection.csv" -delete
ls -la "$O"
