#!/bin/bash
# round 4: planned units with both halves of the workgroup on the one tile (MADQP_CHOL_MID_BOTH): correctness, A/B, stamps
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py -x -q -m gpu > gpurun_out/r4_t19_tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r4_t19_tests.log; tail -5 gpurun_out/r4_t19_tests.log
B="--no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr --no-kernel-timers"
run() { python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo -n "nx5000 both=1: "; run --nx 5000 --m 2000 --steps 40 --warmup 10 $B
  echo -n "nx5000 both=0: "; MADQP_CHOL_MID_BOTH=0 run --nx 5000 --m 2000 --steps 40 --warmup 10 $B
done
for nx in 3000 4000 6000 8000 10000; do
  echo -n "nx$nx both=1: "; run --nx $nx --m $((nx*2/5)) --steps 20 --warmup 5 $B
  echo -n "nx$nx both=0: "; MADQP_CHOL_MID_BOTH=0 run --nx $nx --m $((nx*2/5)) --steps 20 --warmup 5 $B
done
for both in 1 0; do echo "== both=$both"; MADQP_CHOL_MID_BOTH=$both tools/mid_probe 5000 | cut -c1-100 | sed -n '4,8p;30,32p'; done
echo finished
