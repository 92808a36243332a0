set -x
ROOT=$(pwd)
export PROF_DIR=/tmp/prof; mkdir -p $PROF_DIR $ROOT/gpurun_out/r03_out
bash tools/profile_bench.sh r03 --steps 6 --warmup 2 --no-second-ncorr --no-batch-extra
python tools/summarize_pmc.py r03 50000 20000 "--steps 6 --warmup 2 --no-second-ncorr --no-batch-extra" > /dev/null
python tools/analyze_gemm_trace.py "$PROF_DIR/prof_r03_stats/*/*_kernel_trace.csv" 50000 > gpurun_out/r03_out/r03_gemm_launch_breakdown.txt 2>&1
cp profiles/r03_pmc_summary.json profiles/r03_bench50k_kernel_stats.csv gpurun_out/r03_out/
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof/c2db -- python3 $ROOT/bench.py --nx 5000 --m 2000 --steps 30 --warmup 3 --no-kernel-timers --no-cpu-baseline --no-second-ncorr --no-batch-extra > /tmp/prof/c2db.log 2>&1
cd $ROOT; python tools/trace_summary.py /tmp/prof/c2db/*/*results.db > gpurun_out/r03_out/r03_c2_trace_summary.txt 2>&1
tail -5 gpurun_out/r03_out/r03_gemm_launch_breakdown.txt; head -3 gpurun_out/r03_out/r03_c2_trace_summary.txt; du -sh gpurun_out/r03_out
