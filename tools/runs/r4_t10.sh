#!/bin/bash
# round 4: two panels per trailing pass in the mid-size schedule (MADQP_CHOL_MID_TWO): correctness, then A/B timing
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py -x -q -m gpu > gpurun_out/r4_t10_tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r4_t10_tests.log; tail -3 gpurun_out/r4_t10_tests.log
[ $rc -ne 0 ] && exit $rc
B="--no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr --no-kernel-timers"
run() { python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo -n "nx5000 two=1: "; run --nx 5000 --m 2000 --steps 40 --warmup 10 $B
  echo -n "nx5000 two=0: "; MADQP_CHOL_MID_TWO=0 run --nx 5000 --m 2000 --steps 40 --warmup 10 $B
done
for nx in 2000 3000 8000 10000; do
  echo -n "nx$nx two=1: "; run --nx $nx --m $((nx*2/5)) --steps 20 --warmup 5 $B
  echo -n "nx$nx two=0: "; MADQP_CHOL_MID_TWO=0 run --nx $nx --m $((nx*2/5)) --steps 20 --warmup 5 $B
done
echo finished
