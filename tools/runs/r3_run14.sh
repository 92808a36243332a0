run() { python bench.py --kkt distributed --steps 5 --warmup 1 --no-cpu-baseline --no-second-ncorr --no-batch-extra --profile-all 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); nf=d['kkt_factor_solve_ms']['factorizations']; print('$1', round(d['ms_per_step'],1), {k:(round(v['ms']/nf,1),v['launches']//nf) for k,v in d['roofline']['split'].items()})"; }
run new; run new
cp madqp_jl_amd/libmadqp_hip_old.so madqp_jl_amd/libmadqp_hip.so
run old; run old
