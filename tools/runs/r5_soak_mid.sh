#!/bin/bash
# round 5: the random soak at orders 300 .. 1700 (3 .. 14 blocks of 128: mid-size factorisation, backward sweep on U = L',
# the fused per-variable passes on random bound patterns), all drivers against the oracle
mkdir -p gpurun_out
timeout -k 10 1000 python tests/soak_random.py --count 60 --seed0 250000 --nmin 300 --nmax 1700 > gpurun_out/r5_soak_mid.log 2>&1; echo "drivers rc=$?"; tail -3 gpurun_out/r5_soak_mid.log
grep -h MISMATCH gpurun_out/r5_soak_mid.log | head -20
