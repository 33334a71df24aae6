#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/variants
{
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_domain.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -15
GM_LIB_PATH=$L/lib_epi2_st.so python tools/sys_stamps.py 2>&1 | tail -30
for rep in 1 2; do bash tools/ab.sh "--workload target --steps 20 --warmup 5" base epi2 epi3; done
bash tools/ab.sh "--workload c2 --steps 100 --warmup 10" base epi2 epi3
} > gpurun_out/r4_ab3.log 2>&1
cat gpurun_out/r4_ab3.log
