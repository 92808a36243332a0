#!/bin/bash
# round 5: body_fused queued ahead of its read-backs (MADQP_MPC_AHEAD) -- tests, then A/B timing at the mid sizes
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_augmented.py -x -q -m gpu > gpurun_out/r5_t6_tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r5_t6_tests.log; tail -5 gpurun_out/r5_t6_tests.log
[ $rc -ne 0 ] && exit 1
B="--steps 60 --warmup 10 --no-cpu-baseline --no-batch-extra --no-second-ncorr --no-kernel-timers"
run() { timeout -k 10 300 python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], (d.get('whole_solve') or {}).get('iterations_per_s'))"; }
for rep in 1 2; do
echo -n "nx5000 ahead: "; run --nx 5000 --m 2000 $B
echo -n "nx5000 wait: "; MADQP_MPC_AHEAD=0 run --nx 5000 --m 2000 $B
done
for n in 1000 3000 8000; do
echo -n "nx$n ahead: "; run --nx $n --m $((n*2/5)) $B
echo -n "nx$n wait: "; MADQP_MPC_AHEAD=0 run --nx $n --m $((n*2/5)) $B
done
