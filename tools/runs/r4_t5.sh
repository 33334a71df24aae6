#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/variants
{
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_domain.py tests/test_gpu_fullsize.py tests/test_gpu_train.py -x -q -s 2>&1 | tail -25
echo "== parity ps1"; GM_LIB_PATH=$L/lib_ps1.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
for rep in 1 2; do bash tools/ab.sh "--workload target --steps 20 --warmup 5" ps0 ps1; done
bash tools/ab.sh "--workload c2 --steps 100 --warmup 10" ps0 ps1
} > gpurun_out/r4_t5.log 2>&1
cat gpurun_out/r4_t5.log
