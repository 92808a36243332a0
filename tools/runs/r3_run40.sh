set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dense.py tests/test_gpu_dist2d.py -x -q 2>&1 | tail -2
for v in 1 0 1 0; do MADQP_CHOL_PP=$v python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-second-ncorr --no-batch-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('Cmain pp=$v', round(d['value'],4), round(d['ms_per_step'],1), round(d['roofline']['frac'],4))"; done
