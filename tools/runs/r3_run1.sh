set -x
python -m pytest tests/test_gpu_dist2d.py tests/test_gpu_dense.py -x -q -m gpu > gpurun_out/r3_t1.log 2>&1 || { tail -30 gpurun_out/r3_t1.log; exit 1; }
tail -3 gpurun_out/r3_t1.log
python bench.py --kkt distributed --steps 4 --warmup 1 --no-cpu-baseline --no-second-ncorr > gpurun_out/r3_g1_free0.json 2> gpurun_out/r3_g1_free0.err && 
MADQP_DIST_FREE_SLOTS=16 python bench.py --kkt distributed --steps 4 --warmup 1 --no-cpu-baseline --no-second-ncorr > gpurun_out/r3_g1_free16.json 2> gpurun_out/r3_g1_free16.err &&
MADQP_DIST_FREE_SLOTS=32 python bench.py --kkt distributed --steps 4 --warmup 1 --no-cpu-baseline --no-second-ncorr > gpurun_out/r3_g1_free32.json 2> gpurun_out/r3_g1_free32.err &&
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-second-ncorr > gpurun_out/r3_local.json 2> gpurun_out/r3_local.err
python - <<'PY'
import json
for f in ("r3_g1_free0","r3_g1_free16","r3_g1_free32","r3_local"):
    try:
        d=json.load(open(f"gpurun_out/{f}.json")); print(f, d["ms_per_step"], d["kkt_factor_solve_ms"])
    except Exception as e: print(f, "ERR", e)
PY
