#!/bin/bash
mkdir -p gpurun_out
tools/mid_probe 5000 | sed -n "1p;3p;12p;22p;32p;40p"
python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py -x -q -m gpu > gpurun_out/r4_t11_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_t11_tests.log; tail -3 gpurun_out/r4_t11_tests.log
B="--no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr --no-kernel-timers"
run() { python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo -n "nx5000 dsyrk=1: "; run --nx 5000 --m 2000 --steps 40 --warmup 10 $B
  echo -n "nx5000 dsyrk=0: "; MADQP_CHOL_MID_DSYRK=0 run --nx 5000 --m 2000 --steps 40 --warmup 10 $B
done
for nx in 3000 8000; do
  echo -n "nx$nx dsyrk=1: "; run --nx $nx --m $((nx*2/5)) --steps 20 --warmup 5 $B
  echo -n "nx$nx dsyrk=0: "; MADQP_CHOL_MID_DSYRK=0 run --nx $nx --m $((nx*2/5)) --steps 20 --warmup 5 $B
done
