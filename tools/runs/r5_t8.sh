#!/bin/bash
# round 5: sweeps over pairs of blocks on L and U = L' (MADQP_SWEEP_PAIR) -- tests, then A/B timing at the mid sizes
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py tests/test_gpu_augmented.py -x -q -m gpu > gpurun_out/r5_t8_tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r5_t8_tests.log; tail -5 gpurun_out/r5_t8_tests.log
[ $rc -ne 0 ] && exit 1
B="--steps 60 --warmup 10 --no-cpu-baseline --no-batch-extra --no-second-ncorr --no-whole-solve"
run() { timeout -k 10 300 python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kkt_factor_solve_ms'].get('solve_trsv'))"; }
for rep in 1 2; do
echo -n "nx5000 pair: "; run --nx 5000 --m 2000 $B
echo -n "nx5000 single: "; MADQP_SWEEP_PAIR=0 run --nx 5000 --m 2000 $B
done
for n in 1000 3000 8000; do
echo -n "nx$n pair: "; run --nx $n --m $((n*2/5)) $B
echo -n "nx$n single: "; MADQP_SWEEP_PAIR=0 run --nx $n --m $((n*2/5)) $B
done
