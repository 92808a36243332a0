set -x
python __graft_entry__.py smoke 2>&1 | tail -2
python -m pytest tests -x -q -m gpu > gpurun_out/r3_full2.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3_full2.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r3_final_default.json 2> gpurun_out/r3_final_default.err; echo "bench rc=$?"; tail -4 gpurun_out/r3_final_default.err
