#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_dataset.py -x -q -m gpu > gpurun_out/train_tests.log 2>&1
echo "exit $?" >> gpurun_out/train_tests.log
tail -8 gpurun_out/train_tests.log
grep -q "exit 0" gpurun_out/train_tests.log || exit 1
python tools/bench_train.py --steps 20 --no-cpu-baseline > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err || exit 1
cat gpurun_out/bench_train.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trainprof2 -- python3 $R/tools/bench_train.py --steps 10 --no-cpu-baseline > $R/gpurun_out/trainprof2.log 2>&1
cd $R
f=$(find gpurun_out/trainprof2 -name "*kernel_stats.csv" | head -1)
head -16 "$f" | cut -c1-150
