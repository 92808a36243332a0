set -x
export MADQP_DIST_BACKEND=gloo MADQP_DIST_SHARE_DEVICE=1
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-second-ncorr --no-batch-extra --driver python > gpurun_out/r3_reh_n1.json 2> gpurun_out/r3_reh_n1.err
for N in 2 4; do
timeout -k 10 700 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2960$N bench.py --gpus $N --steps 2 --warmup 1 --no-independent-leg > gpurun_out/r3_reh_full_n$N.json 2> gpurun_out/r3_reh_full_n$N.err; echo "N=$N rc=$?"; tail -2 gpurun_out/r3_reh_full_n$N.err
done
python - <<'PY'
import json
for f in ("r3_reh_n1","r3_reh_full_n2","r3_reh_full_n4"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1]); print(f, d["scaling"], round(d["ms_per_step"],1), d["config"]["parallelism"], d.get("last_trace"), d.get("distributed",{}).get("rank0_matrix_bytes"))
    except Exception as e: print(f,"ERR",e)
PY
