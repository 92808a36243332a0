for t in 0 0.5 1.0 2.0; do
MADQP_GEMM_TAIL=$t python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-second-ncorr --no-batch-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tail',$t, round(d['ms_per_step'],1), {k:round(v['ms']/d['kkt_factor_solve_ms']['factorizations'],1) for k,v in d['roofline']['split'].items()})"
done
