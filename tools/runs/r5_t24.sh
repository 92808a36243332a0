#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_solver.py -x -q -m gpu -k "fused_iteration" > gpurun_out/r5_t24_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_t24_tests.log; tail -25 gpurun_out/r5_t24_tests.log
