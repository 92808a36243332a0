#!/bin/bash
# round 4: timeline of one n_x = 5000 iteration (where the idle time sits); repeat of the small-n A/B of the two-panel mode
mkdir -p gpurun_out
B="--no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr --no-kernel-timers"
run() { python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
for nx in 2000 3000; do
  echo -n "nx$nx two=0: "; MADQP_CHOL_MID_TWO=0 run --nx $nx --m $((nx*2/5)) --steps 40 --warmup 10 $B
  echo -n "nx$nx two=1: "; run --nx $nx --m $((nx*2/5)) --steps 40 --warmup 10 $B
done
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -- python3 /root/repo/bench.py --nx 5000 --m 2000 --steps 20 --warmup 5 $B > /dev/null 2>&1
python3 /root/repo/tools/timeline.py /tmp/prof_tl/*/*_kernel_trace.csv > /root/repo/gpurun_out/r4_t13_timeline.txt
python3 /root/repo/tools/trace_summary.py /tmp/prof_tl/*/*_kernel_trace.csv > /root/repo/gpurun_out/r4_t13_summary.txt 2>&1
echo finished
