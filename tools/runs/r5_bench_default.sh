#!/bin/bash
# round 5: the driver's command, wall time and the one JSON line
mkdir -p gpurun_out
t0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_default_line.json 2> gpurun_out/r5_bench_default.err
echo "rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5_bench_default_line.json"))
print(d["metric"]); print("value", d["value"], "ms", d["ms_per_step"], "ref-def", d.get("value_reference_definition"))
print("roofline", d["roofline"]["frac"], "hbm", d.get("roofline_hbm",{}).get("frac"), d["kkt_factor_solve_ms"])
print("c2", json.dumps(d["extras"]["c2_nx5000_m2000"])[:900])
print("batch", d["extras"]["batch_1024x512x256"].get("value"))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["value_source"])
print("whole", d["whole_solve"]["iterations_per_s"], d["whole_solve"]["iter"])
PY
