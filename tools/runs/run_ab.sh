#!/bin/bash
# A/B of library variants on one box: tools/runs/run_ab.sh "tag1 tag2" (libraries variants/lib_<tag>.so, see tools/ab.sh)
set -o pipefail
mkdir -p gpurun_out
TAGS=${1:-"base new"}
{
for rep in 1 2; do bash tools/ab.sh "--workload c4 --steps 8 --warmup 2" $TAGS; done
bash tools/ab.sh "--workload target --steps 20 --warmup 5" $TAGS
bash tools/ab.sh "--workload c2 --steps 100 --warmup 10" $TAGS
} > gpurun_out/ab.log 2>&1
cat gpurun_out/ab.log
