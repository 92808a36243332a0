set -e
python -m pytest tests/test_gpu_dist2d.py -x -q 2>&1 | tail -2
cp madqp_jl_amd/libmadqp_hip.so /tmp/new.so; cp madqp_jl_amd/libmadqp_hip_old.so /tmp/old.so
run() { python bench.py --kkt $2 --steps 5 --warmup 1 --no-cpu-baseline --no-second-ncorr --no-batch-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2', round(d['ms_per_step'],1))"; }
for v in new old new old; do cp /tmp/$v.so madqp_jl_amd/libmadqp_hip.so; run $v distributed; done
cp /tmp/new.so madqp_jl_amd/libmadqp_hip.so; run new local
