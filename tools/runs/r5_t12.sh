#!/bin/bash
# round 5: backward sweep on U = L' with the forward kernel (MADQP_SWEEP_UPPER) -- tests, A/B timing, kernel durations
mkdir -p gpurun_out /tmp/prof
timeout -k 10 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py tests/test_gpu_augmented.py -x -q -m gpu > gpurun_out/r5_t12_tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r5_t12_tests.log; tail -5 gpurun_out/r5_t12_tests.log
[ $rc -ne 0 ] && exit 1
B="--steps 60 --warmup 10 --no-cpu-baseline --no-batch-extra --no-second-ncorr --no-whole-solve --no-kernel-timers"
run() { timeout -k 10 300 python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
echo -n "nx5000 upper: "; run --nx 5000 --m 2000 $B
echo -n "nx5000 lower: "; MADQP_SWEEP_UPPER=0 run --nx 5000 --m 2000 $B
done
for n in 1000 3000 8000; do
echo -n "nx$n upper: "; run --nx $n --m $((n*2/5)) $B
echo -n "nx$n lower: "; MADQP_SWEEP_UPPER=0 run --nx $n --m $((n*2/5)) $B
done
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof/pp -- python3 $ROOT/bench.py --nx 5000 --m 2000 --steps 30 --warmup 3 --no-kernel-timers --no-cpu-baseline --no-second-ncorr --no-batch-extra --no-whole-solve > /tmp/prof/pp.log 2>&1
python3 $ROOT/tools/timeline.py /tmp/prof/pp/*/*results.db > $ROOT/gpurun_out/r5_t12_timeline.txt 2>&1
grep -n "trsv_\|sweep_prep\|lower_to_upper" $ROOT/gpurun_out/r5_t12_timeline.txt | head -12
