#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
bash tools/ab.sh "--workload c4 --steps 8 --warmup 2" base ring2 ring4
bash tools/ab.sh "--workload target --steps 20 --warmup 5" base ring2 ring4 base ring2
bash tools/ab.sh "--workload c2 --steps 100 --warmup 10" base ring2 ring4
} > gpurun_out/ab4.log 2>&1
cat gpurun_out/ab4.log
