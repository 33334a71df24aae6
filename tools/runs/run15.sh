cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
for L in ep2 ep1; do
GM_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_$L.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r3_t15_$L.log 2>&1 || { tail -40 gpurun_out/r3_t15_$L.log; echo TESTS FAILED $L; exit 1; }
tail -2 gpurun_out/r3_t15_$L.log
done
bash tools/ab.sh "--workload target" ep4 ep3 ep2 ep1 ep4 ep3 ep2 ep1 > gpurun_out/r3_ab15.log 2>&1; cat gpurun_out/r3_ab15.log
GM_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_stamps.so timeout -k 10 200 python tools/sys_stamps.py 2>&1 | grep -v amdgpu | grep -E "tick \(|barrier arrival"
