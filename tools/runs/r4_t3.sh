#!/bin/bash
# round 4: mid-size look-ahead correctness + timing, then the measured device / CPU-distance ratios per parity case
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py -x -q -m gpu > gpurun_out/r4_t3_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_t3_tests.log; tail -3 gpurun_out/r4_t3_tests.log
for rep in 1 2; do for la in 1 0; do
  MADQP_CHOL_MID_LOOKAHEAD=$la python bench.py --nx 5000 --m 2000 --steps 40 --warmup 10 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr --no-kernel-timers 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lookahead=$la nx5000', d['value'], d['ms_per_step'])"
done; done
for la in 1 0; do for nx in 3000 8000; do
  MADQP_CHOL_MID_LOOKAHEAD=$la python bench.py --nx $nx --m $((nx*2/5)) --steps 30 --warmup 5 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr --no-kernel-timers 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lookahead=$la nx$nx', d['value'], d['ms_per_step'])"
done; done
timeout -k 10 900 python tests/parity_table.py --out gpurun_out/r4_parity_ratios_sub16.json --refine 0,1 --soak-count 150 2> gpurun_out/r4_t3.log
echo rc=$?
tail -5 gpurun_out/r4_t3.log
