#!/bin/bash
# GPU box: L2 locality experiment on the GEMM core (SYRK n=24576, k=8192) under rocprofv3 PMC.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export PROBE_ONE=1
for cfg in "8 8 1" "8 8 0" "16 4 1" "4 16 1" "64 1 1" "1 64 1" "4 4 1" "16 16 1" "2 32 1"; do
  set -- $cfg
  export MADQP_GEMM_PATCH_M=$1 MADQP_GEMM_PATCH_N=$2 MADQP_GEMM_XCD=$3
  rm -rf $OUT/l2exp
  timeout -k 5 90 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/l2exp -- $ROOT/tools/gemm_probe > $OUT/l2exp.log 2>&1
  python3 - "$cfg" $OUT/l2exp <<'PY'
import csv, glob, sys
cfg, d = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*/*_counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "gemm_tn" in r["Kernel_Name"]]
last = max(r["Dispatch_Id"] for r in rows)
v = {r["Counter_Name"]: float(r["Counter_Value"]) for r in rows if r["Dispatch_Id"] == last}
ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6 for r in rows if r["Dispatch_Id"] == last][0]
print(f"patch/xcd {cfg}: {ms:.1f} ms  fetch {v['FETCH_SIZE']*2048/1e9:.1f} GB", flush=True)
PY
  grep TFLOP $OUT/l2exp.log
done
