#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
for rep in 1 2; do
bash tools/ab.sh "--workload target --steps 20 --warmup 5" base nr1 nr2
done
bash tools/ab.sh "--workload c4 --steps 8 --warmup 2" base nr1 nr2
bash tools/ab.sh "--workload c2 --steps 100 --warmup 10" base nr1
} > gpurun_out/ab_node.log 2>&1
cat gpurun_out/ab_node.log
