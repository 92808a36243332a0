#!/bin/bash
# round 4: batched GEMM dealing whole problems to XCDs (MADQP_GEMM_BATCH_XCD): correctness, then A/B on configs[3]
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_batched.py -x -q -m gpu > gpurun_out/r4_t18_tests.log 2>&1
rc=$?; echo "tests rc=$rc" >> gpurun_out/r4_t18_tests.log; tail -3 gpurun_out/r4_t18_tests.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
  echo -n "xcd=1: "; python tools/bench_batch.py --batch 1024 --repeats 3 2>/dev/null | tail -1 | cut -c1-300
  echo -n "xcd=0: "; MADQP_GEMM_BATCH_XCD=0 python tools/bench_batch.py --batch 1024 --repeats 3 2>/dev/null | tail -1 | cut -c1-300
done
