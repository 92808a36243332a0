#!/bin/bash
# round 5: fused per-variable passes of the condensed mode (MADQP_KKT_FUSE) -- tests, then A/B timing
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_solver.py tests/test_gpu_kernels.py tests/test_gpu_sparse.py tests/test_gpu_augmented.py tests/test_gpu_soak.py -x -q -m gpu > gpurun_out/r5_t13_tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r5_t13_tests.log; tail -5 gpurun_out/r5_t13_tests.log
[ $rc -ne 0 ] && exit 1
B="--steps 60 --warmup 10 --no-cpu-baseline --no-batch-extra --no-second-ncorr --no-whole-solve --no-kernel-timers"
run() { timeout -k 10 300 python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
echo -n "nx5000 fused: "; run --nx 5000 --m 2000 $B
echo -n "nx5000 separate: "; MADQP_KKT_FUSE=0 run --nx 5000 --m 2000 $B
done
for n in 1000 3000 8000; do
echo -n "nx$n fused: "; run --nx $n --m $((n*2/5)) $B
echo -n "nx$n separate: "; MADQP_KKT_FUSE=0 run --nx $n --m $((n*2/5)) $B
done
