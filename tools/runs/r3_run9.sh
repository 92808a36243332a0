set -x
python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r3_t9.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/r3_t9.log
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-batch-extra 2>/dev/null > gpurun_out/r3_symv_main.json
MADQP_SYMV_MIN=100000000 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-batch-extra 2>/dev/null > gpurun_out/r3_nosymv_main.json
python bench.py --nx 5000 --m 2000 --steps 40 --warmup 5 --no-cpu-baseline --no-second-ncorr --no-kernel-timers --no-batch-extra 2>/dev/null > gpurun_out/r3_symv_c2.json
MADQP_SYMV_MIN=100000000 python bench.py --nx 5000 --m 2000 --steps 40 --warmup 5 --no-cpu-baseline --no-second-ncorr --no-kernel-timers --no-batch-extra 2>/dev/null > gpurun_out/r3_nosymv_c2.json
python - <<'PY'
import json
for f in ("r3_symv_main","r3_nosymv_main","r3_symv_c2","r3_nosymv_c2"):
    d=json.load(open(f"gpurun_out/{f}.json")); print(f, round(d["value"],4), round(d["ms_per_step"],3), d.get("max_ncorr_0",{}).get("ms_per_step"))
PY
