set -e
for p in potf2_probe; do echo "== $p"; ./tools/$p | tail -12; done
python -m pytest tests/test_gpu_dense.py -x -q 2>&1 | tail -2
