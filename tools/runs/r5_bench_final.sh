#!/bin/bash
# round 5: the driver's command on the final tree, then the n_x = 5000 kernel trace without the queue-ahead (the profiler's
# per-launch cost makes the host the bottleneck when it queues ahead: the busy fraction of that trace says nothing)
bash tools/runs/r5_bench_default.sh
ROOT=$(pwd); mkdir -p /tmp/prof; cd /tmp; export TMPDIR=/tmp
MADQP_MPC_AHEAD=0 timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof/c2w -- python3 $ROOT/bench.py --nx 5000 --m 2000 --steps 30 --warmup 3 --no-kernel-timers --no-cpu-baseline --no-second-ncorr --no-batch-extra --no-whole-solve > /tmp/prof/c2w.log 2>&1
cd $ROOT; python tools/trace_summary.py /tmp/prof/c2w/*/*results.db > gpurun_out/r05_c2_trace_summary_waiting.txt 2>&1; head -4 gpurun_out/r05_c2_trace_summary_waiting.txt
