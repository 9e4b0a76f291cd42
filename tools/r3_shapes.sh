# split permille and batches in flight on the shapes whose scan is not small beside the walk (configs[2], configs[4])
mkdir -p gpurun_out; : > gpurun_out/r3_shapes2.log
for W in grouping-1B-pq16-nc993127-nsubc64-opq-pruning clustered-grouping-1B-pq16-nc993127-nsubc64-opq-pruning clustered-1B-pq16-nc993127-nprobe32; do
  for SP in 780 700 600; do
    IVFHNSW_SPLIT=$SP timeout -k 10 280 python bench.py --workload $W --no-cpu-baseline --no-secondary --in-flight 2 --sustain-s 0 > gpurun_out/shape.json 2>/dev/null || exit 1
    python3 -c "
import json
o=json.loads([l for l in open('gpurun_out/shape.json') if l.startswith('{')][-1])
print('$W split $SP: value', o['value'], 'ms', o['ms_per_step'], 'one_part', o['one_part']['queries_per_s'] if o.get('one_part') else None, o['one_part']['stage_ms_per_step'] if o.get('one_part') else None, 'pipelined', o['pipelined']['queries_per_s'])" >> gpurun_out/r3_shapes2.log
  done
done
cat gpurun_out/r3_shapes2.log
