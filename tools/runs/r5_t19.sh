#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_solver.py -x -q -m gpu -k "synthetic_vs_oracle" > gpurun_out/r5_t19_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_t19_tests.log; tail -8 gpurun_out/r5_t19_tests.log
