#!/bin/bash
# GPU box: kernel stats + FETCH_SIZE + WRITE_SIZE passes of the 1024 x (512, 256) batch (BASELINE configs[3])
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout -k 10 400 rocprofv3 "$@" --output-format csv -d $OUT/prof_r03batch_$name -- python3 $ROOT/tools/bench_batch.py --batch 1024 --repeats 1 > $OUT/prof_r03batch_$name.log 2>&1
  echo "$name rc=$?"; tail -1 $OUT/prof_r03batch_$name.log | cut -c1-200; }
run stats --kernel-trace --stats && run fetch --pmc FETCH_SIZE --kernel-trace && run write --pmc WRITE_SIZE --kernel-trace
