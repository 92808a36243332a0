set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r33_tests.log 2>&1 || { tail -30 gpurun_out/r33_tests.log; exit 1; }
tail -2 gpurun_out/r33_tests.log
