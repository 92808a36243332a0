#!/bin/bash
# round 4: factor-only diagonal kernels + one inversion launch (MADQP_CHOL_LITE): correctness and A/B timing
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py tests/test_gpu_dist2d.py tests/test_gpu_augmented.py tests/test_gpu_batched.py tests/test_gpu_random.py -x -q -m gpu > gpurun_out/r4_t9_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_t9_tests.log; tail -3 gpurun_out/r4_t9_tests.log
B="--no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr --no-kernel-timers"
run() { python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo -n "nx5000 lite=1: "; run --nx 5000 --m 2000 --steps 40 --warmup 10 $B
  echo -n "nx5000 lite=0: "; MADQP_CHOL_LITE=0 run --nx 5000 --m 2000 --steps 40 --warmup 10 $B
done
for nx in 3000 8000; do
  echo -n "nx$nx lite=1: "; run --nx $nx --m $((nx*2/5)) --steps 20 --warmup 5 $B
  echo -n "nx$nx lite=0: "; MADQP_CHOL_LITE=0 run --nx $nx --m $((nx*2/5)) --steps 20 --warmup 5 $B
done
echo -n "cmain lite=1: "; run --steps 4 --warmup 1 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr
echo -n "cmain lite=0: "; MADQP_CHOL_LITE=0 run --steps 4 --warmup 1 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr
