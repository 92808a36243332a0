#!/bin/bash
# round 5: the parity ratio table (device traces against the LAPACK oracle in units of the ensemble floor) per sweep mode:
#   $1 = nrm16 (default) | sub16 | inv ; $2 = refine settings (default "0")
mkdir -p gpurun_out
MODE=${1:-nrm16}
MADQP_SWEEP_DIAG=$MODE timeout -k 10 1100 python tests/parity_table.py --out gpurun_out/r5_parity_ratios_$MODE.json --refine ${2:-0} --soak-count 150 2> gpurun_out/r5_pt_$MODE.log | tail -30
echo "rc=${PIPESTATUS[0]}"
