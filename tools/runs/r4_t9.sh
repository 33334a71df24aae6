#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/variants
{
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for rep in 1 2; do bash tools/ab.sh "--workload target --steps 20 --warmup 5" noenc p113 p220 p120 p210 p111; done
} > gpurun_out/r4_t9.log 2>&1
cat gpurun_out/r4_t9.log
