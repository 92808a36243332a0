#!/bin/bash
# GPU box: L2 locality experiment on the GEMM core (SYRK n=24576, k=8192) under rocprofv3 PMC.
# Variants are selected with the MADQP_GEMM_* environment knobs of gemm_f64.hip.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export PROBE_ONE=1
for cfg in "0 8 1 0 50000" "64 8 1 0 50000" "32 8 1 0 50000" "16 8 1 0 50000" "8 8 1 0 50000"; do
  set -- $cfg
  export MADQP_GEMM_SEG_ROUNDS=$1 MADQP_GEMM_PATCH_M=8 MADQP_GEMM_PATCH_N=$2 MADQP_GEMM_XCD=$3 MADQP_GEMM_PACE=$4 PROBE_N=$5
  rm -rf $OUT/l2exp
  timeout -k 5 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/l2exp -- $ROOT/tools/gemm_probe > $OUT/l2exp.log 2>&1
  python3 - "$cfg" $OUT/l2exp <<'PY'
import csv, glob, sys
cfg, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*/*_counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm_tn" in r["Kernel_Name"]]
# the probe runs the assembly 4 times (1 warm-up + 3 timed): average per run over all its launches
tot = sum(float(r["Counter_Value"]) for r in rows) / 4.0
v = {"FETCH_SIZE": tot}
ms = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rows) / 4.0
print(f"seg_rounds patch xcd pace n = {cfg}: {ms:.1f} ms  fetch {v['FETCH_SIZE']*2048/1e9:.1f} GB", flush=True)
PY
  grep TFLOP $OUT/l2exp.log
done
