#!/bin/bash
# round 5: kernel timelines of one iteration at n_x = 5000 with and without the queued-ahead body
ROOT=$(pwd); mkdir -p gpurun_out
mkdir -p /tmp/prof; cd /tmp; export TMPDIR=/tmp
for mode in 1 0; do
MADQP_MPC_AHEAD=$mode timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof/tl$mode -- python3 $ROOT/bench.py --nx 5000 --m 2000 --steps 30 --warmup 3 --no-kernel-timers --no-cpu-baseline --no-second-ncorr --no-batch-extra --no-whole-solve > /tmp/prof/tl$mode.log 2>&1
echo "mode $mode rc=$?"; tail -1 /tmp/prof/tl$mode.log | cut -c1-200
python3 $ROOT/tools/timeline.py /tmp/prof/tl$mode/*/*results.db > $ROOT/gpurun_out/r5_t7_timeline_ahead$mode.txt 2>&1
python3 $ROOT/tools/trace_summary.py /tmp/prof/tl$mode/*/*results.db > $ROOT/gpurun_out/r5_t7_summary_ahead$mode.txt 2>&1
done
