#!/bin/bash
# round 4: planned visits of the trailing columns (MADQP_CHOL_MID_LAZY): correctness, A/B timing, per-step durations
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py -x -q -m gpu > gpurun_out/r4_t14_tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r4_t14_tests.log; tail -3 gpurun_out/r4_t14_tests.log
[ $rc -ne 0 ] && exit $rc
B="--no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr --no-kernel-timers"
run() { python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo -n "nx5000 lazy=1: "; run --nx 5000 --m 2000 --steps 40 --warmup 10 $B
  echo -n "nx5000 pairs=0: "; MADQP_CHOL_MID_PAIRS=0 run --nx 5000 --m 2000 --steps 40 --warmup 10 $B
done
for nx in 2000 3000 4000 6000 8000 10000; do
  echo -n "nx$nx lazy=1: "; run --nx $nx --m $((nx*2/5)) --steps 20 --warmup 5 $B
  echo -n "nx$nx pairs=0: "; MADQP_CHOL_MID_PAIRS=0 run --nx $nx --m $((nx*2/5)) --steps 20 --warmup 5 $B
done
cd /tmp && export TMPDIR=/tmp
for nx in 5000 8000; do
  rm -rf /tmp/prof_l
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_l -- python3 /root/repo/bench.py --nx $nx --m $((nx*2/5)) --steps 20 --warmup 5 $B > /dev/null 2>&1
  echo "== steps nx=$nx lazy"; python3 /root/repo/tools/mid_steps.py /tmp/prof_l/*/*_kernel_trace.csv
done
echo finished
