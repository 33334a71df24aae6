#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
for rep in 1 2; do bash tools/ab.sh "--workload target --steps 20 --warmup 5" p113 p003 p013 p103 p223 p213 p123; done
} > gpurun_out/r4_t10.log 2>&1
cat gpurun_out/r4_t10.log
