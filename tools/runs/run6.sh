cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
fail() { echo "FAILED: $1"; exit 1; }
for L in d e1; do
  GM_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_$L.so timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r3_t6_$L.log 2>&1 || { tail -40 gpurun_out/r3_t6_$L.log; fail "tests $L"; }
  tail -2 gpurun_out/r3_t6_$L.log
done
bash tools/ab.sh "--workload target" c113 d e1 e1p0 e1p321 c113 d e1 e1p0 e1p321 > gpurun_out/r3_ab6.log 2>&1
cat gpurun_out/r3_ab6.log
grep -q "Memory access fault" gpurun_out/r3_ab6.log && fail "fault in ab"
for L in stampsd stampse; do
  echo "--- $L"
  GM_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_$L.so timeout -k 10 200 python tools/sys_stamps.py > gpurun_out/r3_$L.log 2>&1 || { tail -5 gpurun_out/r3_$L.log; fail "stamps $L"; }
  cat gpurun_out/r3_$L.log
done
