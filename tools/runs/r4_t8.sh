#!/bin/bash
# round 4: tail split A/B, then the parity ratio table with the refined sweeps (MADQP_SWEEP_REFINE=1)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r4_t8_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_t8_tests.log; tail -3 gpurun_out/r4_t8_tests.log
B="--no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr --no-kernel-timers"
run() { python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  echo -n "nx5000 tail=1: "; run --nx 5000 --m 2000 --steps 40 --warmup 10 $B
  echo -n "nx5000 tail=0: "; MADQP_GEMM_TAILSPLIT=0 run --nx 5000 --m 2000 --steps 40 --warmup 10 $B
done
for nx in 3000 8000 12000 20000; do
  echo -n "nx$nx tail=1: "; run --nx $nx --m $((nx*2/5)) --steps 12 --warmup 3 $B
  echo -n "nx$nx tail=0: "; MADQP_GEMM_TAILSPLIT=0 run --nx $nx --m $((nx*2/5)) --steps 12 --warmup 3 $B
done
echo -n "cmain tail=1: "; run --steps 4 --warmup 1 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr
echo -n "cmain tail=0: "; MADQP_GEMM_TAILSPLIT=0 run --steps 4 --warmup 1 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr
MADQP_SWEEP_REFINE=1 timeout -k 10 800 python tests/parity_table.py --out gpurun_out/r4_parity_ratios_sweep_refine.json --refine 0 --soak-count 150 2> gpurun_out/r4_t8.log | tail -22
