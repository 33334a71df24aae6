#!/bin/bash
# round 4, A/B 1: role-0 rotation
set -o pipefail
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/variants
{
GM_LIB_PATH=$L/lib_rot.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
GM_LIB_PATH=$L/lib_rot_st.so python tools/sys_stamps.py 2>&1 | tail -30
for rep in 1 2; do bash tools/ab.sh "--workload target --steps 20 --warmup 5" base rot; done
bash tools/ab.sh "--workload c2 --steps 100 --warmup 10" base rot
} > gpurun_out/r4_ab1.log 2>&1
cat gpurun_out/r4_ab1.log
