# round 4: the driver's bench command on the final tree
mkdir -p gpurun_out/r04_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_out/r04_bench_default_line_final.json 2> gpurun_out/r04_out/r04_bench_default_final.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r04_out/r04_bench_default_line_final.json'))
print({k:d[k] for k in ('value','ms_per_step','n_gpus')}); print(d['roofline']['frac'], d['roofline'].get('mfma_busy_counter')); print(d['whole_solve']['iterations_per_s'], d['whole_solve']['iter']); print(d['cpu_baseline']['value'], d['extras']['batch_1024x512x256']['value'])"
