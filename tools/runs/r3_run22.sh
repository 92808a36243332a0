set -e
for v in 1024 512 1024 512; do MADQP_BATCH_WIDE_MAX=$v python tools/bench_batch.py --batch 1024 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3 wide_max=$v', round(d['value']), d['all_seconds'], d['solved'])"; done
