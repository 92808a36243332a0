# round 4 final check: what the driver does (pytest -m gpu, smoke, bench) + configs[1] line + batch profile
set -x
mkdir -p gpurun_out/r04_out
python -m pytest tests -x -q -m gpu > gpurun_out/r04_out/r04_gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04_out/r04_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_out/r04_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r04_out/r04_smoke.log
python bench.py --nx 5000 --m 2000 --steps 40 --warmup 10 --no-cpu-baseline --no-batch-extra --no-kernel-timers > gpurun_out/r04_out/r04_c2_bench_line.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r04_out/r04_c2_bench_line.json')); print('c2', d['value'], d['ms_per_step'], d['max_ncorr_0']['value'], d['whole_solve']['iterations_per_s'], d['whole_solve']['iter'])"
ROOT=$(pwd); OUT=$ROOT/gpurun_out; cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout -k 10 700 rocprofv3 "$@" --output-format csv -d $OUT/prof_r04batch_$name -- python3 $ROOT/tools/bench_batch.py --batch 1024 --repeats 1 > $OUT/prof_r04batch_$name.log 2>&1
  echo "$name rc=$?"; }
run stats --kernel-trace --stats && run fetch --pmc FETCH_SIZE --kernel-trace && run write --pmc WRITE_SIZE --kernel-trace
cd $ROOT; python tools/summarize_batch_prof.py r04batch > /dev/null 2>&1; cp profiles/r04batch_summary.* gpurun_out/r04_out/; head -8 gpurun_out/r04_out/r04batch_summary.txt
