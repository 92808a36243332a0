#!/bin/bash
# round 5: sweeps with the right-hand-side prefetch and one barrier fewer per hop -- tests, then timing
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py tests/test_gpu_augmented.py tests/test_gpu_sparse.py tests/test_gpu_dist2d.py -x -q -m gpu > gpurun_out/r5_t4_tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r5_t4_tests.log; tail -5 gpurun_out/r5_t4_tests.log
[ $rc -ne 0 ] && exit 1
B="--steps 60 --warmup 10 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr --no-kernel-timers"
run() { timeout -k 10 300 python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
echo -n "nx5000 nrm16: "; run --nx 5000 --m 2000 $B
echo -n "nx5000 inv: "; MADQP_SWEEP_DIAG=inv run --nx 5000 --m 2000 $B
done
echo -n "nx3000 nrm16: "; run --nx 3000 --m 1200 $B
C="--steps 6 --warmup 2 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr"
for mode in nrm16 inv; do
echo -n "cmain $mode: "; MADQP_SWEEP_DIAG=$mode timeout -k 10 400 python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kkt_factor_solve_ms']['solve_trsv'], d.get('roofline_hbm',{}).get('avg_launch_ms'))"
done
