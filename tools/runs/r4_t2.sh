#!/bin/bash
# round 4: block-substitution panel kernel -- correctness (dense, solver, batched, distributed leaves) and A/B timing
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py tests/test_gpu_batched.py tests/test_gpu_dist2d.py -x -q -m gpu > gpurun_out/r4_t2_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_t2_tests.log
tail -4 gpurun_out/r4_t2_tests.log
for rep in 1 2; do
for mode in sub16 inv; do
  MADQP_CHOL_PANEL=$mode python bench.py --nx 5000 --m 2000 --steps 40 --warmup 10 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr --no-kernel-timers 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode nx5000', d['value'], d['ms_per_step'])"
done; done
for mode in sub16 inv; do
  MADQP_CHOL_PANEL=$mode python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode cmain', d['value'], d['ms_per_step'], d['kkt_factor_solve_ms'], d['roofline']['split'].get('potrf_trsm'))"
done
python tools/bench_batch.py 2>&1 | tail -3
MADQP_CHOL_PANEL=inv python tools/bench_batch.py 2>&1 | tail -3
