#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
for rep in 1 2; do bash tools/ab.sh "--workload target --steps 20 --warmup 5" a0 a1 a2 a3 a4 a16 a31; done
} > gpurun_out/r4_abl.log 2>&1
cat gpurun_out/r4_abl.log
