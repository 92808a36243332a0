set -x
python -m pytest tests/test_gpu_dist2d.py -x -q -m gpu > gpurun_out/r3_t2.log 2>&1 || { tail -40 gpurun_out/r3_t2.log; exit 1; }
tail -3 gpurun_out/r3_t2.log
for nb in 1024 1536 2048; do
python bench.py --kkt distributed --panel-width $nb --steps 4 --warmup 1 --no-cpu-baseline --no-second-ncorr > gpurun_out/r3_g1_nb$nb.json 2> gpurun_out/r3_g1_nb$nb.err || exit 1
done
python - <<'PY'
import json
for f in ("r3_g1_nb1024","r3_g1_nb1536","r3_g1_nb2048"):
    try:
        d=json.load(open(f"gpurun_out/{f}.json")); print(f, round(d["ms_per_step"],1), {k:(round(v["ms"]/4,1),v["launches"]//4) for k,v in d["roofline"]["split"].items()})
    except Exception as e: print(f, "ERR", e)
PY
