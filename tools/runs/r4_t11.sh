#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
for rep in 1 2; do bash tools/ab.sh "--workload c2 --steps 100 --warmup 10" prev short; done
bash tools/ab.sh "--workload target --steps 20 --warmup 5" prev short
bash tools/ab.sh "--workload c3 --steps 20 --warmup 5" prev short
} > gpurun_out/r4_t11.log 2>&1
cat gpurun_out/r4_t11.log
