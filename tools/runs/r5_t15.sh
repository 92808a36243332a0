#!/bin/bash
# round 5: U = L' copied on a side stream -- tests, then timing
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py tests/test_gpu_soak.py -x -q -m gpu > gpurun_out/r5_t15_tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r5_t15_tests.log; tail -5 gpurun_out/r5_t15_tests.log
[ $rc -ne 0 ] && exit 1
B="--steps 60 --warmup 10 --no-cpu-baseline --no-batch-extra --no-second-ncorr --no-whole-solve --no-kernel-timers"
run() { timeout -k 10 300 python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
echo -n "nx5000: "; run --nx 5000 --m 2000 $B
done
for n in 1000 3000 8000; do
echo -n "nx$n: "; run --nx $n --m $((n*2/5)) $B
done
