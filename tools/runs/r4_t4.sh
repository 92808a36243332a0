#!/bin/bash
# round 4: refined diagonal step of the sweeps, look-ahead v2, single-launch reductions: correctness + A/B timings
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py tests/test_gpu_kernels.py tests/test_gpu_batched.py tests/test_gpu_dist2d.py -x -q -m gpu > gpurun_out/r4_t4_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_t4_tests.log; tail -3 gpurun_out/r4_t4_tests.log
B="--steps 40 --warmup 10 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr --no-kernel-timers"
run() { python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  echo -n "nx5000 default: "; run --nx 5000 --m 2000 $B
  echo -n "nx5000 lookahead=0: "; MADQP_CHOL_MID_LOOKAHEAD=0 run --nx 5000 --m 2000 $B
  echo -n "nx5000 fused_red=0: "; MADQP_FUSED_REDUCTIONS=0 run --nx 5000 --m 2000 $B
done
echo -n "nx3000 default: "; run --nx 3000 --m 1200 $B
echo -n "nx3000 lookahead=0: "; MADQP_CHOL_MID_LOOKAHEAD=0 run --nx 3000 --m 1200 $B
echo -n "nx8000 default: "; run --nx 8000 --m 3200 --steps 20 --warmup 5 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr --no-kernel-timers
echo -n "nx8000 lookahead=0: "; MADQP_CHOL_MID_LOOKAHEAD=0 run --nx 8000 --m 3200 --steps 20 --warmup 5 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr --no-kernel-timers
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cmain', d['value'], d['ms_per_step'], d['kkt_factor_solve_ms'], d.get('roofline_hbm',{}).get('avg_launch_ms'))"
python tools/bench_batch.py --batch 1024 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch1024', d['value'])"
