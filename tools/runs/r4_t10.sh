#!/bin/bash
# round 4: compacted retry rounds of the batched engine: correctness + A/B
mkdir -p gpurun_out
python -m pytest tests/test_gpu_batched.py tests/test_gpu_random.py tests/test_gpu_soak.py -x -q -m gpu > gpurun_out/r4_t10_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_t10_tests.log; tail -3 gpurun_out/r4_t10_tests.log
for rep in 1 2 3; do
python tools/bench_batch.py --batch 1024 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('compact=1 batch1024', d['value'])"
MADQP_BATCH_RETRY_COMPACT=0 python tools/bench_batch.py --batch 1024 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('compact=0 batch1024', d['value'])"
done
python tools/bench_batch.py --batch 128 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('compact=1 batch128', d['value'])"
MADQP_BATCH_RETRY_COMPACT=0 python tools/bench_batch.py --batch 128 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('compact=0 batch128', d['value'])"
