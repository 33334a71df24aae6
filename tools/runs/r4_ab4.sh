#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/variants
{
for v in r10 r01 r11; do echo "== parity $v"; GM_LIB_PATH=$L/lib_$v.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2; done
GM_LIB_PATH=$L/lib_r11_st.so python tools/sys_stamps.py 2>&1 | tail -30
for rep in 1 2; do bash tools/ab.sh "--workload target --steps 20 --warmup 5" r00 r10 r01 r11; done
bash tools/ab.sh "--workload c2 --steps 100 --warmup 10" r00 r10 r01 r11
} > gpurun_out/r4_ab4.log 2>&1
cat gpurun_out/r4_ab4.log
