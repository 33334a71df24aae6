#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/variants
{
echo "== parity f1"; GM_LIB_PATH=$L/lib_f1.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
GM_LIB_PATH=$L/lib_f1.so timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py -x -q -k "target_size or c3_size or two_kernel" 2>&1 | tail -3
GM_LIB_PATH=$L/lib_f1_st.so timeout -k 10 120 python tools/sys_stamps.py 2>&1 | tail -36
for rep in 1 2; do timeout -k 10 200 bash tools/ab.sh "--workload target --steps 20 --warmup 5" f0 f1; done
timeout -k 10 200 bash tools/ab.sh "--workload c2 --steps 100 --warmup 10" f0 f1
} > gpurun_out/r4_t7.log 2>&1
cat gpurun_out/r4_t7.log
