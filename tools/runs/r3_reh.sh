set -x
export MADQP_DIST_BACKEND=gloo MADQP_DIST_SHARE_DEVICE=1
for N in 2 4; do
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2960$N bench.py --gpus $N --steps 2 --warmup 1 --nx 20000 --m 8000 --no-independent-leg > gpurun_out/r3_reh_n$N.json 2> gpurun_out/r3_reh_n$N.err; echo "N=$N rc=$?"; tail -2 gpurun_out/r3_reh_n$N.err; cut -c1-300 gpurun_out/r3_reh_n$N.json
done
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29609 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r3_reh_full_n2.json 2> gpurun_out/r3_reh_full_n2.err; echo "full N=2 rc=$?"; tail -2 gpurun_out/r3_reh_full_n2.err
python - <<'PY'
import json
for f in ("r3_reh_n2","r3_reh_n4","r3_reh_full_n2"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1]); print(f, d["scaling"], round(d["ms_per_step"],1), d["config"]["parallelism"], d.get("last_trace"), d.get("independent_qps",{}).get("ms_per_step"))
    except Exception as e: print(f,"ERR",e)
PY
