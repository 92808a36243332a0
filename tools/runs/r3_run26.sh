set -e

cp madqp_jl_amd/libmadqp_hip.so /tmp/new.so; cp madqp_jl_amd/libmadqp_hip_old.so /tmp/old.so
for v in new old new old new old; do cp /tmp/$v.so madqp_jl_amd/libmadqp_hip.so; python tools/bench_batch.py --batch 1024 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3 $v', round(d['value']), d['all_seconds'])"; done
cp /tmp/new.so madqp_jl_amd/libmadqp_hip.so
