#!/bin/bash
# round 4: the diagonal kernel stores its factor column block by column block beside the pivot chain: correctness + timing
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py tests/test_gpu_batched.py tests/test_gpu_dist2d.py tests/test_gpu_augmented.py -x -q -m gpu > gpurun_out/r4_t21_tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r4_t21_tests.log; tail -4 gpurun_out/r4_t21_tests.log
[ $rc -ne 0 ] && exit $rc
tools/mid_probe 2048 | tail -10
B="--no-cpu-baseline --no-batch-extra --no-whole-solve --no-kernel-timers"
run() { python bench.py $* 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['max_ncorr_0']['value'])"; }
for rep in 1 2 3; do
  echo -n "nx5000: "; run --nx 5000 --m 2000 --steps 40 --warmup 10 $B
done
python tools/bench_batch.py --batch 1024 --repeats 3 2>/dev/null | tail -1 | cut -c1-120
echo finished
