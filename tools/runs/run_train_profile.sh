#!/bin/bash
# Training bench (tools/bench_train.py) + its rocprofv3 kernel statistics.
set -o pipefail
mkdir -p gpurun_out
python tools/bench_train.py --steps 20 --no-cpu-baseline > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err || exit 1
cat gpurun_out/bench_train.json
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trainprof -- python3 $R/tools/bench_train.py --steps 10 --no-cpu-baseline > $R/gpurun_out/trainprof.log 2>&1
cd $R
f=$(find gpurun_out/trainprof -name "*kernel_stats.csv" | head -1)
head -18 "$f" | cut -c1-150
