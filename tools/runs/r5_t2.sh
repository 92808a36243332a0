#!/bin/bash
# round 5: the whole GPU suite with the trace comparison of the soak ON (default library: nrm16 sweeps + AUTO refinement)
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r5_t2_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_t2_tests.log; tail -15 gpurun_out/r5_t2_tests.log
