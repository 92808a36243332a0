# round 5: the committed profiles -- C-main kernel stats + PMC passes (panel GEMM counter), GEMM launch breakdown, n_x = 5000 trace
set -x
ROOT=$(pwd)
export PROF_DIR=/tmp/prof; mkdir -p $PROF_DIR $ROOT/gpurun_out/r05_out
bash tools/profile_bench.sh r05 --steps 6 --warmup 2 --no-second-ncorr --no-batch-extra --no-whole-solve
python tools/summarize_pmc.py r05 50000 20000 "--steps 6 --warmup 2 --no-second-ncorr --no-batch-extra --no-whole-solve" > /dev/null
python tools/analyze_gemm_trace.py "$PROF_DIR/prof_r05_stats/*/*_kernel_trace.csv" 50000 > gpurun_out/r05_out/r05_gemm_launch_breakdown.txt 2>&1
cp profiles/r05_pmc_summary.json profiles/r05_bench50k_kernel_stats.csv gpurun_out/r05_out/
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof/c2db -- python3 $ROOT/bench.py --nx 5000 --m 2000 --steps 30 --warmup 3 --no-kernel-timers --no-cpu-baseline --no-second-ncorr --no-batch-extra --no-whole-solve > /tmp/prof/c2db.log 2>&1
cd $ROOT; python tools/trace_summary.py /tmp/prof/c2db/*/*results.db > gpurun_out/r05_out/r05_c2_trace_summary.txt 2>&1
tail -5 gpurun_out/r05_out/r05_gemm_launch_breakdown.txt; head -12 gpurun_out/r05_out/r05_c2_trace_summary.txt; du -sh gpurun_out/r05_out
python - <<'PY'
import json
d=json.load(open("profiles/r05_pmc_summary.json")); print(json.dumps(d.get("mfma_busy"), indent=1))
PY
# batch engine: kernel stats + FETCH / WRITE passes condensed by tools/summarize_batch_prof.py (reads gpurun_out/prof_<tag>_*)
cd /tmp
for pass in stats fetch write; do
  case $pass in stats) A="--kernel-trace --stats";; fetch) A="--pmc FETCH_SIZE --kernel-trace";; write) A="--pmc WRITE_SIZE --kernel-trace";; esac
  timeout -k 10 300 rocprofv3 $A --output-format csv -d /tmp/prof/prof_r05batch_$pass -- python3 $ROOT/tools/bench_batch.py --batch 1024 --repeats 1 > /tmp/prof/batch_$pass.log 2>&1
  echo "batch $pass rc=$?"
done
cd $ROOT
mkdir -p gpurun_out/tmpb && rm -rf gpurun_out/prof_r05batch_* && for pass in stats fetch write; do mkdir -p gpurun_out/prof_r05batch_$pass/x; cp /tmp/prof/prof_r05batch_$pass/*/*_kernel_stats.csv /tmp/prof/prof_r05batch_$pass/*/*_counter_collection.csv gpurun_out/prof_r05batch_$pass/x/ 2>/dev/null; done
python tools/summarize_batch_prof.py r05batch > gpurun_out/r05_out/r05batch.log 2>&1; tail -3 gpurun_out/r05_out/r05batch.log
cp profiles/r05batch_summary.* gpurun_out/r05_out/ 2>/dev/null
rm -rf gpurun_out/prof_r05batch_* gpurun_out/tmpb
ls gpurun_out/r05_out
