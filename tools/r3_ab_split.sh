for SP in 780 600 780 600; do
  IVFHNSW_SPLIT=$SP timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --no-one-part --in-flight 1 --sustain-s 1.0 > gpurun_out/ab.json 2>/dev/null || exit 1
  python3 -c "
import json
o=json.loads([l for l in open('gpurun_out/ab.json') if l.startswith('{')][-1])
print('split $SP', o['value'], o['ms_per_step'], o['sustained']['queries_per_s'], o['batch_split']['part_queries'], o['roofline']['frac'])"
done
