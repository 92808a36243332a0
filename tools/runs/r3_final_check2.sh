# GPU box: the round's last check -- smoke, the whole GPU suite, the n_x = 5000 rate, the default bench line, the profiles
set -e
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -1
python -m pytest tests -x -q -m gpu > gpurun_out/final_tests.log 2>&1 || { tail -30 gpurun_out/final_tests.log; exit 1; }
tail -2 gpurun_out/final_tests.log
for i in 1 2 3; do python bench.py --nx 5000 --ncon 2000 --steps 40 --warmup 5 --no-cpu-baseline --no-second-ncorr --no-kernel-timers --no-batch-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n_x=5000', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms')"; done
python bench.py --steps 10 --warmup 2 > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err
tail -1 gpurun_out/final_bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('default line:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_hbm']['frac'], d['cpu_baseline']['value'], d['extras']['batch_1024x512x256']['value'])"
bash tools/runs/r3_final_prof.sh > gpurun_out/final_prof.log 2>&1
tail -3 gpurun_out/final_prof.log
