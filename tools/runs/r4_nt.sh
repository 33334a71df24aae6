#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
for rep in 1 2; do bash tools/ab.sh "--workload target --steps 20 --warmup 5" n0 n1 n2 n4 n5 n7; done
bash tools/ab.sh "--workload c2 --steps 100 --warmup 10" n0 n7
} > gpurun_out/r4_nt.log 2>&1
cat gpurun_out/r4_nt.log
