#!/bin/bash
# round 5, first GPU call: the new diagonal step of the sweeps (unit block substitution) -- dense tests, solver tests,
# then timing A/B of the three sweep modes at n_x = 5000 and at C-main
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/r5_t1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_t1_tests.log; tail -5 gpurun_out/r5_t1_tests.log
B="--steps 60 --warmup 10 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr --no-kernel-timers"
run() { python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for mode in nrm16 sub16 inv; do
echo -n "nx5000 $mode: "; MADQP_SWEEP_DIAG=$mode run --nx 5000 --m 2000 $B
done
done
for mode in nrm16 inv; do
echo -n "nx3000 $mode: "; MADQP_SWEEP_DIAG=$mode run --nx 3000 --m 1200 $B
done
C="--steps 6 --warmup 2 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr"
for mode in nrm16 inv; do
echo -n "cmain $mode: "; MADQP_SWEEP_DIAG=$mode python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kkt_factor_solve_ms'], d.get('roofline_hbm',{}).get('avg_launch_ms'))"
done
