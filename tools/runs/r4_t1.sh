#!/bin/bash
# round 4, first GPU call: the distributed tests after the staircase stores / symv / p2p changes, bench launcher on the GPU
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dist2d.py tests/test_gpu_bench.py tests/test_gpu_julia_replay.py -x -q -m gpu > gpurun_out/r4_t1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4_t1_tests.log
tail -5 gpurun_out/r4_t1_tests.log
python bench.py --nx 5000 --m 2000 --steps 20 --warmup 5 --no-cpu-baseline --no-batch-extra > gpurun_out/r4_t1_bench5k.json 2> gpurun_out/r4_t1_bench5k.err
echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4_t1_bench5k.json"))
print({k:d[k] for k in ("value","ms_per_step","n_gpus","n_gpus_requested")}, d.get("whole_solve"))
PY
