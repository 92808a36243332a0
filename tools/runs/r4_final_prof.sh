# round 4: the committed profiles -- C-main kernel stats + PMC passes (panel GEMM counter), GEMM launch breakdown, n_x = 5000 trace
set -x
ROOT=$(pwd)
export PROF_DIR=/tmp/prof; mkdir -p $PROF_DIR $ROOT/gpurun_out/r04_out
bash tools/profile_bench.sh r04 --steps 6 --warmup 2 --no-second-ncorr --no-batch-extra --no-whole-solve
python tools/summarize_pmc.py r04 50000 20000 "--steps 6 --warmup 2 --no-second-ncorr --no-batch-extra --no-whole-solve" > /dev/null
python tools/analyze_gemm_trace.py "$PROF_DIR/prof_r04_stats/*/*_kernel_trace.csv" 50000 > gpurun_out/r04_out/r04_gemm_launch_breakdown.txt 2>&1
cp profiles/r04_pmc_summary.json profiles/r04_bench50k_kernel_stats.csv gpurun_out/r04_out/
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof/c2db -- python3 $ROOT/bench.py --nx 5000 --m 2000 --steps 30 --warmup 3 --no-kernel-timers --no-cpu-baseline --no-second-ncorr --no-batch-extra --no-whole-solve > /tmp/prof/c2db.log 2>&1
cd $ROOT; python tools/trace_summary.py /tmp/prof/c2db/*/*results.db > gpurun_out/r04_out/r04_c2_trace_summary.txt 2>&1
tail -5 gpurun_out/r04_out/r04_gemm_launch_breakdown.txt; head -12 gpurun_out/r04_out/r04_c2_trace_summary.txt; du -sh gpurun_out/r04_out
python - <<'PY'
import json
d=json.load(open("profiles/r04_pmc_summary.json")); print(json.dumps(d.get("mfma_busy"), indent=1))
PY
