# round 4: the driver's command, the GEMM launch breakdown with the updated trace analysis, the batch profile
set -x
ROOT=$(pwd); mkdir -p gpurun_out/r04_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_out/r04_bench_default_line.json 2> gpurun_out/r04_out/r04_bench_default.err
echo "bench rc=$?"; tail -3 gpurun_out/r04_out/r04_bench_default.err
cd /tmp; export TMPDIR=/tmp; mkdir -p /tmp/prof
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/st -- python3 $ROOT/bench.py --no-cpu-baseline --steps 6 --warmup 2 --no-second-ncorr --no-batch-extra --no-whole-solve > /tmp/prof/st.log 2>&1
cd $ROOT; python tools/analyze_gemm_trace.py "/tmp/prof/st/*/*_kernel_trace.csv" 50000 > gpurun_out/r04_out/r04_gemm_launch_breakdown.txt 2>&1
head -8 gpurun_out/r04_out/r04_gemm_launch_breakdown.txt
sed -i 's/r03batch/r04batch/g' tools/runs/r3_batch_prof.sh && bash tools/runs/r3_batch_prof.sh && python tools/summarize_batch_prof.py r04batch > /dev/null 2>&1; ls profiles | grep r04batch; cp profiles/r04batch* gpurun_out/r04_out/ 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r04_out/r04_bench_default_line.json'))
print({k:d[k] for k in ('value','ms_per_step','n_gpus','n_gpus_requested')}); print(d['roofline']['frac'], d['roofline']['mfma_busy_counter']); print(d['whole_solve']); print(d['cpu_baseline']['value'], d['extras']['batch_1024x512x256']['value'])"
