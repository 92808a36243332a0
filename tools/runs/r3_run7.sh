set -x
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/prof_r03c2 -- python3 $ROOT/bench.py --nx 5000 --m 2000 --steps 20 --warmup 3 --no-cpu-baseline --no-second-ncorr --no-kernel-timers > $ROOT/gpurun_out/r3_c2_trace.log 2>&1
cd $ROOT; ls gpurun_out/prof_r03c2/*/ | head; python tools/timeline.py gpurun_out/prof_r03c2/*/*_kernel_trace.csv > gpurun_out/r3_c2_timeline.txt; tail -3 gpurun_out/r3_c2_timeline.txt
