#!/bin/bash
# round 4, final tree: the parity ratio table of the default library (refine_steps 0 and 1); $1 = refine: the refined sweeps instead
mkdir -p gpurun_out
if [ "$1" = "refine" ]; then
  MADQP_SWEEP_REFINE=1 timeout -k 10 1000 python tests/parity_table.py --out gpurun_out/r4_parity_ratios_sweep_refine.json --refine 0 --soak-count 150 2> gpurun_out/r4_pt_refine.log | tail -24
else
  timeout -k 10 1050 python tests/parity_table.py --out gpurun_out/r4_parity_ratios_default.json --refine 0,1 --soak-count 150 2> gpurun_out/r4_pt_default.log | tail -48
fi
echo "rc=${PIPESTATUS[0]}"
