#!/bin/bash
mkdir -p gpurun_out
MADQP_SWEEP_STAMPS=1 timeout -k 10 300 python bench.py --nx 5000 --m 2000 --steps 20 --warmup 5 --no-cpu-baseline --no-batch-extra --no-second-ncorr --no-whole-solve --no-kernel-timers 2> gpurun_out/r5_t11_stamps.txt | cut -c1-100
grep -c STAMP gpurun_out/r5_t11_stamps.txt
