#!/bin/bash
B="--no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr --no-kernel-timers"
run() { python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for cap in 254 230 200 170 140; do
  echo -n "nx5000 cap=$cap: "; MADQP_CHOL_MID_CAP=$cap run --nx 5000 --m 2000 --steps 40 --warmup 10 $B
done
done
