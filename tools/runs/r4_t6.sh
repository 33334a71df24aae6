#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/variants
{
GM_LIB_PATH=$L/lib_cur_st.so python tools/sys_stamps.py 2>&1 | tail -36
for rep in 1 2; do bash tools/ab.sh "--workload target --steps 20 --warmup 5" e2 e1; done
timeout -k 10 900 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -5
} > gpurun_out/r4_t6.log 2>&1
cat gpurun_out/r4_t6.log
