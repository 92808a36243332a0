set -x
python -m pytest tests/test_gpu_dist2d.py tests/test_gpu_julia_replay.py -x -q -m gpu 2>&1 | tail -3
python bench.py --kkt distributed --steps 4 --warmup 1 --no-cpu-baseline --no-second-ncorr --no-batch-extra > gpurun_out/r3_g1_nb1024.json 2> gpurun_out/r3_g1_nb1024.err || exit 1
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-second-ncorr --no-batch-extra > gpurun_out/r3_local.json 2> gpurun_out/r3_local.err
MADQP_DIST_FORCE_RCCL=1 python bench.py --kkt distributed --steps 4 --warmup 1 --no-cpu-baseline --no-second-ncorr --no-batch-extra > gpurun_out/r3_g1_force.json 2> gpurun_out/r3_g1_force.err; tail -2 gpurun_out/r3_g1_force.err
python - <<'PY'
import json
for f in ("r3_g1_nb1024","r3_local","r3_g1_force"):
    try:
        d=json.load(open(f"gpurun_out/{f}.json")); print(f, round(d["ms_per_step"],1), {k:(round(v["ms"]/4,1),v["launches"]//4) for k,v in d["roofline"]["split"].items()})
    except Exception as e: print(f, "ERR", e)
PY
