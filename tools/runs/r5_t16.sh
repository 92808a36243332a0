#!/bin/bash
# round 5: the partial sums of a block's other jobs requested together -- tests, then the sweeps at C-main and n_x = 20000
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py -x -q -m gpu > gpurun_out/r5_t16_tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r5_t16_tests.log; tail -5 gpurun_out/r5_t16_tests.log
[ $rc -ne 0 ] && exit 1
C="--steps 6 --warmup 2 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr"
echo -n "cmain: "; timeout -k 10 500 python bench.py $C 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kkt_factor_solve_ms']['solve_trsv'], json.dumps(d.get('roofline_hbm',{}))[:300])"
echo -n "nx20000: "; timeout -k 10 300 python bench.py --nx 20000 --m 8000 --steps 10 --warmup 2 --no-cpu-baseline --no-batch-extra --no-whole-solve --no-second-ncorr 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kkt_factor_solve_ms']['solve_trsv'])"
