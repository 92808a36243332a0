#!/bin/bash
# round 5: pair sweeps, second version -- dense + solver tests, then A/B timing at n_x = 5000 and the kernel durations
mkdir -p gpurun_out /tmp/prof
timeout -k 10 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py -x -q -m gpu > gpurun_out/r5_t10_tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r5_t10_tests.log; tail -5 gpurun_out/r5_t10_tests.log
[ $rc -ne 0 ] && exit 1
B="--steps 60 --warmup 10 --no-cpu-baseline --no-batch-extra --no-second-ncorr --no-whole-solve --no-kernel-timers"
run() { timeout -k 10 300 python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
echo -n "nx5000 pair: "; run --nx 5000 --m 2000 $B
echo -n "nx5000 single: "; MADQP_SWEEP_PAIR=0 run --nx 5000 --m 2000 $B
done
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof/pp -- python3 $ROOT/bench.py --nx 5000 --m 2000 --steps 30 --warmup 3 --no-kernel-timers --no-cpu-baseline --no-second-ncorr --no-batch-extra --no-whole-solve > /tmp/prof/pp.log 2>&1
python3 $ROOT/tools/timeline.py /tmp/prof/pp/*/*results.db > $ROOT/gpurun_out/r5_t10_timeline.txt 2>&1
grep -n "trsv_pair\|sweep_prep\|lower_to_upper" $ROOT/gpurun_out/r5_t10_timeline.txt | head -12
