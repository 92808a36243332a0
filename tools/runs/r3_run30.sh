set -e
for ce in 2 4 6 9 2 4 6 9; do python tools/bench_batch.py --batch 1024 --check-every $ce 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3 check_every=$ce', round(d['value']), d['all_seconds'], d['solved'])"; done
