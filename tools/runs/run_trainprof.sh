#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
R=$PWD
python tools/bench_train.py --steps 20 --no-cpu-baseline > gpurun_out/bench_train.json 2> gpurun_out/bench_train.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/trainprof -o tp -- python3 $R/tools/bench_train.py --steps 10 --no-cpu-baseline > $R/gpurun_out/trainprof.log 2>&1
cd $R
f=$(find gpurun_out/trainprof -name "*kernel_stats.csv" | head -1)
head -40 "$f" | cut -c1-200
cat gpurun_out/bench_train.json
