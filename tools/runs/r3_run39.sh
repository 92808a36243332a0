set -e
python -m pytest tests/test_gpu_dense.py -x -q 2>&1 | tail -2
for v in 1 0 1 0 1 0; do MADQP_CHOL_PP=$v python bench.py --nx 5000 --ncon 2000 --steps 40 --warmup 5 --no-cpu-baseline --no-second-ncorr --no-kernel-timers --no-batch-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C2 pp=$v', round(d['value'],1), round(d['ms_per_step'],3))"; done
for n in 3000 8000; do for v in 1 0; do MADQP_CHOL_PP=$v python bench.py --nx $n --ncon $((n*2/5)) --steps 20 --warmup 3 --no-cpu-baseline --no-second-ncorr --no-kernel-timers --no-batch-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('n=$n pp=$v', round(d['value'],1), round(d['ms_per_step'],3))"; done; done
