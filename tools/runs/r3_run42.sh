set -e
./tools/mid_probe 5000 | sed -n 1,5p; ./tools/mid_probe 5000 | sed -n 30,33p
python -m pytest tests/test_gpu_dense.py -x -q 2>&1 | tail -2
