set -x
( time python bench.py ) > gpurun_out/r3_default.json 2> gpurun_out/r3_default.err; echo "bench rc=$?"
tail -5 gpurun_out/r3_default.err
bash tools/runs/r3_vec_pmc.sh
