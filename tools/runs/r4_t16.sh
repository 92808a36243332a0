#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py tests/test_gpu_random.py tests/test_gpu_configs.py -x -q -m gpu -s > gpurun_out/r4_t16_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_t16_tests.log; tail -4 gpurun_out/r4_t16_tests.log; grep -E "^(planned|two_panels|every_panel|gemm_diagonal) " gpurun_out/r4_t16_tests.log
