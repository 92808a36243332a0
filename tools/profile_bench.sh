#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# Usage: [PROF_DIR=/tmp/prof] tools/profile_bench.sh <tag> [bench args...]; writes $PROF_DIR/prof_<tag>_*/ (default gpurun_out/)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${PROF_DIR:-$ROOT/gpurun_out}   # raw traces are large: point PROF_DIR at /tmp and keep only the summaries
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS=("$@")
run() { # name, rocprof args...
  local name=$1; shift
  timeout -k 10 600 rocprofv3 "$@" --output-format csv -d $OUT/prof_${TAG}_$name -- python3 $ROOT/bench.py --no-cpu-baseline "${BENCH_ARGS[@]}" > $OUT/prof_${TAG}_$name.log 2>&1
  echo "$name rc=$?"; grep -o '"value": [0-9.e+-]*' $OUT/prof_${TAG}_$name.log | head -1
}
run stats --kernel-trace --stats
run fetch --pmc FETCH_SIZE --kernel-trace
run write --pmc WRITE_SIZE --kernel-trace
run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace
run tcc --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace
run sq2 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM --kernel-trace
ls $OUT/prof_${TAG}_*/*/ 2>/dev/null | head -40
