#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
for rep in 1 2; do
bash tools/ab.sh "--workload c4 --steps 8 --warmup 2" base lnc
done
bash tools/ab.sh "--workload target --steps 20 --warmup 5" base lnc
} > gpurun_out/ab2.log 2>&1
cat gpurun_out/ab2.log
