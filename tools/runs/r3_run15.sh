set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r15_tests.log 2>&1 && tail -3 gpurun_out/r15_tests.log
for i in 1 2; do python bench.py --nx 5000 --ncon 2000 --steps 40 --warmup 5 --no-cpu-baseline --no-second-ncorr --no-kernel-timers --no-batch-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C2', d['value'], d['ms_per_step'])"; done
for i in 1 2; do python tools/bench_batch.py --batch 1024 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3', d['value'], d['all_seconds'])"; done
