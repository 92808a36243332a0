set -e
python -m pytest tests/test_gpu_batched.py -x -q 2>&1 | tail -2
for i in 1 2 3; do python tools/bench_batch.py --batch 1024 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3', round(d['value']), d['all_seconds'], d['solved'])"; done
