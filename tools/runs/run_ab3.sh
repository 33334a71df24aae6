#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_domain.py -x -q -m gpu 2>&1 | tail -3
bash tools/ab.sh "--workload c4 --steps 8 --warmup 2" base ring4
bash tools/ab.sh "--workload target --steps 20 --warmup 5" base ring4
bash tools/ab.sh "--workload c2 --steps 100 --warmup 10" base ring4
} > gpurun_out/ab3.log 2>&1
cat gpurun_out/ab3.log
