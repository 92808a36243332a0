set -e
python -m pytest tests/test_gpu_batched.py -x -q 2>&1 | tail -2
python tools/runs/r3_run27.py | tail -3
