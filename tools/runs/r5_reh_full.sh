#!/bin/bash
# round 5: rehearsal of `bench.py --gpus 2` at the FULL metric size, two ranks sharing the GPU over gloo (host-staged collectives)
mkdir -p gpurun_out
t0=$(date +%s)
MADQP_DIST_BACKEND=gloo MADQP_DIST_SHARE_DEVICE=1 timeout -k 10 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --extra-timeout 400 > gpurun_out/r5_reh_full_n2.json 2> gpurun_out/r5_reh_full_n2.err
echo "rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_reh_full_n2.json"))
print(d["n_gpus"], d["scaling"], d["value"], d["ms_per_step"], d["comm"], d["distributed"]["grid"], d["independent_qps"]["value"], d["last_trace"])
PY
tail -3 gpurun_out/r5_reh_full_n2.err
