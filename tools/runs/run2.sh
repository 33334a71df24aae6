set -e
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export GM_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_c.so
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r3_t2.log 2>&1 || { tail -30 gpurun_out/r3_t2.log; exit 1; }
tail -3 gpurun_out/r3_t2.log
unset GM_LIB_PATH
bash tools/ab.sh "--workload target" b c c113 c012 c000 b c c113 c012 c000 > gpurun_out/r3_ab2.log 2>&1
cat gpurun_out/r3_ab2.log
GM_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_stamps.so timeout -k 10 200 python tools/sys_stamps.py > gpurun_out/r3_stamps2.log 2>&1
cat gpurun_out/r3_stamps2.log
