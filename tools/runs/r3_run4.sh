set -x
python -m pytest tests/test_gpu_batched.py -x -q -m gpu > gpurun_out/r3_t4.log 2>&1 || { tail -40 gpurun_out/r3_t4.log; exit 1; }
tail -3 gpurun_out/r3_t4.log
MADQP_CHOL_SMALL_MAX=0 python tools/bench_batch.py --batch 1024 > gpurun_out/r3_batch_old.json 2> gpurun_out/r3_batch_old.err
python tools/bench_batch.py --batch 1024 > gpurun_out/r3_batch_new.json 2> gpurun_out/r3_batch_new.err
python tools/bench_batch.py --batch 1024 --profile --repeats 1 > gpurun_out/r3_batch_prof.txt 2>&1
python tools/bench_batch.py --batch 128 > gpurun_out/r3_batch128_new.json 2> gpurun_out/r3_batch128_new.err
cat gpurun_out/r3_batch_old.json gpurun_out/r3_batch_new.json gpurun_out/r3_batch128_new.json; head -3 gpurun_out/r3_batch_prof.txt
