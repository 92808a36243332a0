#!/bin/bash
# round 5: kernel summary at n_x = 5000 with the pair sweeps
ROOT=$(pwd); mkdir -p gpurun_out /tmp/prof
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof/pp -- python3 $ROOT/bench.py --nx 5000 --m 2000 --steps 30 --warmup 3 --no-kernel-timers --no-cpu-baseline --no-second-ncorr --no-batch-extra --no-whole-solve > /tmp/prof/pp.log 2>&1
echo "rc=$?"
python3 $ROOT/tools/trace_summary.py /tmp/prof/pp/*/*results.db > $ROOT/gpurun_out/r5_t9_summary.txt 2>&1
python3 $ROOT/tools/timeline.py /tmp/prof/pp/*/*results.db > $ROOT/gpurun_out/r5_t9_timeline.txt 2>&1
head -14 $ROOT/gpurun_out/r5_t9_summary.txt
