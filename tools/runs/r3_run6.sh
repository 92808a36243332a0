set -x
python -m pytest tests/test_gpu_dense.py tests/test_gpu_solver.py -x -q -m gpu > gpurun_out/r3_t6.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r3_t6.log
for i in 1 2; do
MADQP_GEMM_SPLITK=0 python bench.py --nx 5000 --m 2000 --steps 40 --warmup 5 --no-cpu-baseline --no-second-ncorr --no-kernel-timers > gpurun_out/r3_c2_nosplit_$i.json 2>/dev/null
python bench.py --nx 5000 --m 2000 --steps 40 --warmup 5 --no-cpu-baseline --no-second-ncorr --no-kernel-timers > gpurun_out/r3_c2_split_$i.json 2>/dev/null
done
python - <<'PY'
import json
for f in ("r3_c2_nosplit_1","r3_c2_split_1","r3_c2_nosplit_2","r3_c2_split_2"):
    d=json.load(open(f"gpurun_out/{f}.json")); print(f, round(d["value"],1), round(d["ms_per_step"],3))
PY
