set -e
mkdir -p gpurun_out
./tools/potf2_probe | tail -12
python -m pytest tests -x -q -m gpu > gpurun_out/r37_tests.log 2>&1 || { tail -30 gpurun_out/r37_tests.log; exit 1; }
tail -2 gpurun_out/r37_tests.log
