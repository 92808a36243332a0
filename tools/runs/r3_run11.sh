set -x
python -m pytest tests/test_gpu_dist2d.py -x -q -m gpu 2>&1 | tail -3
for nb in 1024 1280 1536 1792; do
python bench.py --kkt distributed --panel-width $nb --steps 4 --warmup 1 --no-cpu-baseline --no-second-ncorr --no-batch-extra > gpurun_out/r3_g1_nb$nb.json 2> gpurun_out/r3_g1_nb$nb.err || exit 1
done
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-second-ncorr --no-batch-extra > gpurun_out/r3_local.json 2> gpurun_out/r3_local.err
python - <<'PY'
import json
for f in ("r3_g1_nb1024","r3_g1_nb1280","r3_g1_nb1536","r3_g1_nb1792","r3_local"):
    try:
        d=json.load(open(f"gpurun_out/{f}.json")); print(f, round(d["ms_per_step"],1), {k:(round(v["ms"]/4,1),v["launches"]//4) for k,v in d["roofline"]["split"].items()})
    except Exception as e: print(f, "ERR", e)
PY
