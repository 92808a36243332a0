#!/bin/bash
# round 5: the random soak on seeds no test has seen, default library (substituting sweeps + AUTO refinement); second range at the end of the round
mkdir -p gpurun_out
timeout -k 10 500 python tests/soak_random.py --count 600 --seed0 150000 > gpurun_out/r5_soak_drivers.log 2>&1; echo "drivers rc=$?"; tail -2 gpurun_out/r5_soak_drivers.log
timeout -k 10 300 python tests/soak_random.py --count 300 --seed0 160000 --mode sparse > gpurun_out/r5_soak_sparse.log 2>&1; echo "sparse rc=$?"; tail -2 gpurun_out/r5_soak_sparse.log
timeout -k 10 300 python tests/soak_random.py --count 300 --seed0 170000 --mode augmented > gpurun_out/r5_soak_aug.log 2>&1; echo "augmented rc=$?"; tail -2 gpurun_out/r5_soak_aug.log
grep -h MISMATCH gpurun_out/r5_soak_*.log | head -30
