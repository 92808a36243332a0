set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_batched.py -x -q > gpurun_out/r16_tests.log 2>&1 || { tail -30 gpurun_out/r16_tests.log; exit 1; }
tail -3 gpurun_out/r16_tests.log
for v in 1 0 1 0; do MADQP_BATCH_SYMV=$v python tools/bench_batch.py --batch 1024 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3 symv=$v', round(d['value']), d['all_seconds'], d['solved'])"; done
