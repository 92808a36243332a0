# round 4, last check: the driver's steps on the final tree + configs[1] line and trace
set -x
mkdir -p gpurun_out/r04_out
python -m pytest tests -x -q -m gpu > gpurun_out/r04_out/r04_gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_out/r04_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_out/r04_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r04_out/r04_smoke.log
python bench.py --nx 5000 --m 2000 --steps 40 --warmup 10 --no-cpu-baseline --no-batch-extra --no-kernel-timers > gpurun_out/r04_out/r04_c2_bench_line.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r04_out/r04_c2_bench_line.json')); print('c2', d['value'], d['ms_per_step'], d['max_ncorr_0']['value'], d['whole_solve']['iterations_per_s'], d['whole_solve']['iter'])"
ROOT=$(pwd); cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof/c2db /tmp/prof/c2csv; mkdir -p /tmp/prof
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof/c2db -- python3 $ROOT/bench.py --nx 5000 --m 2000 --steps 30 --warmup 3 --no-kernel-timers --no-cpu-baseline --no-second-ncorr --no-batch-extra --no-whole-solve > /tmp/prof/c2db.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof/c2csv -- python3 $ROOT/bench.py --nx 5000 --m 2000 --steps 30 --warmup 3 --no-kernel-timers --no-cpu-baseline --no-second-ncorr --no-batch-extra --no-whole-solve > /tmp/prof/c2csv.log 2>&1
cd $ROOT; python tools/trace_summary.py /tmp/prof/c2db/*/*results.db > gpurun_out/r04_out/r04_c2_trace_summary.txt 2>&1
python tools/mid_steps.py /tmp/prof/c2csv/*/*_kernel_trace.csv > gpurun_out/r04_out/r04_c2_mid_steps.txt 2>&1
head -4 gpurun_out/r04_out/r04_c2_trace_summary.txt; tail -2 gpurun_out/r04_out/r04_c2_mid_steps.txt
python tools/bench_batch.py --batch 1024 --repeats 3 2>/dev/null | tail -1 | cut -c1-200
