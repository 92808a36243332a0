set -e
ROOT=$(pwd); mkdir -p gpurun_out/r03_out /tmp/prof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/dist1x1 -- python3 $ROOT/bench.py --kkt distributed --steps 4 --warmup 1 --no-cpu-baseline --no-second-ncorr --no-batch-extra > /tmp/prof/dist1x1.log 2>&1
cd $ROOT
tail -1 /tmp/prof/dist1x1.log | cut -c1-300
cp /tmp/prof/dist1x1/*/*_kernel_stats.csv gpurun_out/r03_out/r03_dist1x1_kernel_stats.csv
head -12 gpurun_out/r03_out/r03_dist1x1_kernel_stats.csv | cut -c1-160
