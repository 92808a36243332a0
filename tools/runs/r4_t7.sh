#!/bin/bash
# round 4: whole GPU suite, then the parity ratio table of the default library
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r4_t7_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_t7_tests.log; tail -4 gpurun_out/r4_t7_tests.log
timeout -k 10 850 python tests/parity_table.py --out gpurun_out/r4_parity_ratios_default.json --refine 0,1 --soak-count 150 2> gpurun_out/r4_t7.log
echo rc=$?
