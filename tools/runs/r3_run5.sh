set -x
python tests/golden/make_seed9195.py gpurun_out/seed9195_oracle_gpu_box_host.json
python -m pytest tests/test_gpu_dist2d.py tests/test_gpu_augmented.py tests/test_gpu_solver.py -x -q -m gpu > gpurun_out/r3_t5a.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/r3_t5a.log
python -m pytest tests/test_gpu_soak.py -x -q -m gpu -s > gpurun_out/r3_t5b.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/r3_t5b.log
