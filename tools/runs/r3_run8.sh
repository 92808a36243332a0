set -x
python -m pytest tests/test_gpu_batched.py tests/test_gpu_dense.py -x -q -m gpu > gpurun_out/r3_t8.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r3_t8.log
python tools/bench_batch.py --batch 1024 > gpurun_out/r3_batch_xcd.json 2>/dev/null; cut -c1-420 gpurun_out/r3_batch_xcd.json
python tools/bench_batch.py --batch 128 2>/dev/null | cut -c1-160
python tools/bench_batch.py --batch 1024 --profile --repeats 1 2>/dev/null | head -1
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-second-ncorr --no-batch-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
