#!/bin/bash
# round 5: `bench.py --gpus 4` at the FULL metric size on a 2 x 2 grid, four ranks sharing the GPU over gloo; both broadcast forms
mkdir -p gpurun_out
for form in collective p2p; do
t0=$(date +%s)
[ $form = p2p ] && export MADQP_DIST_BCAST=p2p
MADQP_DIST_BACKEND=gloo MADQP_DIST_SHARE_DEVICE=1 timeout -k 10 500 python bench.py --gpus 4 --steps 2 --warmup 1 --no-cpu-baseline --no-independent-leg --extra-timeout 400 > gpurun_out/r5_reh_full_n4_$form.json 2> gpurun_out/r5_reh_full_n4_$form.err
echo "$form rc=$? wall=$(( $(date +%s) - t0 )) s"
python - $form <<'PY'
import json, sys
d=json.load(open(f"gpurun_out/r5_reh_full_n4_{sys.argv[1]}.json"))
print(d["n_gpus"], d["scaling"], d["value"], d["ms_per_step"], d["comm"]["row_comm_size"], d["comm"]["col_comm_size"], d["distributed"]["grid"], d["distributed"]["bytes_broadcast_by_rank0"], d["last_trace"])
PY
done
