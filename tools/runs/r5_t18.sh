#!/bin/bash
# round 5: batched sweeps without the zero half of the inverse images -- tests, then QP/s
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_batched.py tests/test_gpu_soak.py -x -q -m gpu > gpurun_out/r5_t18_tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r5_t18_tests.log; tail -5 gpurun_out/r5_t18_tests.log
[ $rc -ne 0 ] && exit 1
for rep in 1 2 3; do
echo -n "batch1024: "; timeout -k 10 300 python tools/bench_batch.py --batch 1024 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d.get('seconds'), d.get('lock_step_iterations'))"
done
echo -n "batch128: "; timeout -k 10 300 python tools/bench_batch.py --batch 128 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])"
