set -e
python -m pytest tests/test_gpu_dense.py -x -q 2>&1 | tail -2
cp madqp_jl_amd/libmadqp_hip.so /tmp/keep.so
for rep in 1 2 3; do for v in old new; do cp madqp_jl_amd/libvar_$v.so madqp_jl_amd/libmadqp_hip.so
python bench.py --nx 5000 --ncon 2000 --steps 40 --warmup 5 --no-cpu-baseline --no-second-ncorr --no-kernel-timers --no-batch-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C2 $v', round(d['value'],1), round(d['ms_per_step'],3))"
done; done
cp /tmp/keep.so madqp_jl_amd/libmadqp_hip.so
