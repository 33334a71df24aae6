set -e
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export GM_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_d.so
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r3_t3.log 2>&1 || { tail -40 gpurun_out/r3_t3.log; exit 1; }
tail -3 gpurun_out/r3_t3.log
unset GM_LIB_PATH
bash tools/ab.sh "--workload target" c113 d c113 d > gpurun_out/r3_ab3.log 2>&1
cat gpurun_out/r3_ab3.log
GM_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_stamps.so timeout -k 10 200 python tools/sys_stamps.py > gpurun_out/r3_stamps3.log 2>&1
cat gpurun_out/r3_stamps3.log
