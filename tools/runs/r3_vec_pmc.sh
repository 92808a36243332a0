#!/bin/bash
# GPU box: kernel-trace stats + FETCH_SIZE + WRITE_SIZE passes of tools/vec_pmc.py (per-variable kernels at C5 size)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout -k 10 300 rocprofv3 "$@" --output-format csv -d $OUT/prof_r03vec_$name -- python3 $ROOT/tools/vec_pmc.py 100000 40000 > $OUT/prof_r03vec_$name.log 2>&1
  echo "$name rc=$?"; }
run stats --kernel-trace --stats && run fetch --pmc FETCH_SIZE --kernel-trace && run write --pmc WRITE_SIZE --kernel-trace
cd $ROOT && python3 tools/summarize_vec_pmc.py r03vec 100000 40000 > $OUT/r03_vec_pmc_table.txt 2>&1; tail -40 $OUT/r03_vec_pmc_table.txt
