#!/bin/bash
# round 4: the whole GPU suite with the ensemble-floor parity rule, then timing of the default library and of MADQP_SWEEP_REFINE=1
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r4_t5_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_t5_tests.log; tail -5 gpurun_out/r4_t5_tests.log
B="--steps 40 --warmup 10 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr --no-kernel-timers"
run() { python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
echo -n "nx5000 default: "; run --nx 5000 --m 2000 $B
echo -n "nx5000 refine: "; MADQP_SWEEP_REFINE=1 run --nx 5000 --m 2000 $B
python tools/bench_batch.py --batch 1024 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch1024', d['value'])"
