#!/bin/bash
# round 5: the whole GPU suite on the final tree + smoke
mkdir -p gpurun_out
timeout -k 10 1150 python -m pytest tests -x -q -m gpu > gpurun_out/r5_final_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_final_tests.log; tail -6 gpurun_out/r5_final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
