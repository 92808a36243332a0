#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_gpu_dist2d.py -x -q -m gpu -k "kkt" > gpurun_out/r5_t25_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_t25_tests.log; tail -25 gpurun_out/r5_t25_tests.log
