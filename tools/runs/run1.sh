set -e
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
export GM_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_b.so
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t1.log 2>&1 || { tail -30 gpurun_out/r3_t1.log; exit 1; }
tail -3 gpurun_out/r3_t1.log
unset GM_LIB_PATH
bash tools/ab.sh "--workload target" r2 a b r2 a b > gpurun_out/r3_ab1.log 2>&1
cat gpurun_out/r3_ab1.log
bash tools/ab.sh "--workload c2" r2 b > gpurun_out/r3_ab1c2.log 2>&1
cat gpurun_out/r3_ab1c2.log
GM_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_stamps.so timeout -k 10 200 python tools/sys_stamps.py > gpurun_out/r3_stamps1.log 2>&1
cat gpurun_out/r3_stamps1.log
