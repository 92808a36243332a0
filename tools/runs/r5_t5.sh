#!/bin/bash
# round 5: batched engine with the incremental model evaluation -- parity tests, then QP/s with and without it
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_soak.py -x -q -m gpu -k "batched or soak" > gpurun_out/r5_t5_tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r5_t5_tests.log; tail -5 gpurun_out/r5_t5_tests.log
[ $rc -ne 0 ] && exit 1
for rep in 1 2; do
for mode in 1 0; do
echo -n "batch1024 incr=$mode: "; MADQP_BATCH_INCR=$mode timeout -k 10 300 python tools/bench_batch.py --batch 1024 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d.get('seconds'), d.get('lock_step_iterations'))"
done
done
echo -n "batch128 incr=1: "; timeout -k 10 300 python tools/bench_batch.py --batch 128 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])"
echo -n "batch128 incr=0: "; MADQP_BATCH_INCR=0 timeout -k 10 300 python tools/bench_batch.py --batch 128 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])"
