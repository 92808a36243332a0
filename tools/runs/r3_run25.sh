set -e
cp madqp_jl_amd/libmadqp_hip.so /tmp/new.so
for rep in 1 2; do for v in old B C D E; do if [ $v = old ]; then src=madqp_jl_amd/libmadqp_hip_old.so; else src=madqp_jl_amd/libvar_$v.so; fi; cp $src madqp_jl_amd/libmadqp_hip.so; python tools/bench_batch.py --batch 1024 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3 $v', round(d['value']), d['all_seconds'])"; done; done
cp /tmp/new.so madqp_jl_amd/libmadqp_hip.so
