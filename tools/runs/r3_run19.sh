set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r19_tests.log 2>&1 || { tail -30 gpurun_out/r19_tests.log; exit 1; }
tail -2 gpurun_out/r19_tests.log
python bench.py --steps 10 --warmup 2 > gpurun_out/r19_bench_default.json 2> gpurun_out/r19_bench_default.err
python -c "
import json
d=json.loads(open('gpurun_out/r19_bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('roofline_hbm',{}).get('frac'), d['cpu_baseline']['value'], d['extras']['batch_1024x512x256']['value'] if 'extras' in d else None)
"
