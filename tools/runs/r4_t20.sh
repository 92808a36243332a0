#!/bin/bash
# round 4: mu / tau / mu_c stay on the device in the fused iteration (two read-backs instead of three, one per Gondzio trial
# instead of two): correctness, then timing at several sizes
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_kernels.py tests/test_gpu_random.py tests/test_gpu_augmented.py tests/test_gpu_julia_replay.py -x -q -m gpu > gpurun_out/r4_t20_tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r4_t20_tests.log; tail -5 gpurun_out/r4_t20_tests.log
[ $rc -ne 0 ] && exit $rc
B="--no-cpu-baseline --no-batch-extra --no-whole-solve --no-kernel-timers"
run() { python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['max_ncorr_0']['value'])"; }
for rep in 1 2 3; do
  echo -n "nx5000: "; run --nx 5000 --m 2000 --steps 40 --warmup 10 $B
done
for nx in 1000 2000 3000 8000; do
  echo -n "nx$nx: "; run --nx $nx --m $((nx*2/5)) --steps 20 --warmup 5 $B
done
echo finished
