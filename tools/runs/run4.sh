cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export GM_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_e1.so
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r3_t4.log 2>&1 || { tail -40 gpurun_out/r3_t4.log; exit 1; }
tail -3 gpurun_out/r3_t4.log
unset GM_LIB_PATH
bash tools/ab.sh "--workload target" d e1 e1p0 e1p321 d e1 e1p0 e1p321 > gpurun_out/r3_ab4.log 2>&1
cat gpurun_out/r3_ab4.log
echo "--- stamps tool on lib_d (no stamps: expect SystemExit after the forwards)"
GM_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_d.so timeout -k 10 200 python tools/sys_stamps.py 2>&1 | tail -3
echo "--- stamps (variant e1)"
GM_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_stampse.so timeout -k 10 200 python tools/sys_stamps.py > gpurun_out/r3_stamps4.log 2>&1
tail -40 gpurun_out/r3_stamps4.log
