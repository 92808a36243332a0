#!/bin/bash
# round 4: where the two-panel schedule gains and loses -- phase stamps and per-step kernel durations, both modes
mkdir -p gpurun_out
for two in 1 0; do
  echo "== mid_probe 2048 two=$two"; MADQP_CHOL_MID_TWO=$two tools/mid_probe 2048 | cut -c1-110
  echo "== mid_probe 5000 two=$two"; MADQP_CHOL_MID_TWO=$two tools/mid_probe 5000 | cut -c1-110
done
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr --no-kernel-timers"
for nx in 2000 5000; do
for two in 1 0; do
  rm -rf /tmp/prof_$two
  MADQP_CHOL_MID_TWO=$two timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$two -- python3 /root/repo/bench.py --nx $nx --m $((nx*2/5)) --steps 20 --warmup 5 $B > /dev/null 2>&1
  echo "== steps nx=$nx two=$two"; python3 /root/repo/tools/mid_steps.py /tmp/prof_$two/*/*_kernel_trace.csv
done
done
echo finished
