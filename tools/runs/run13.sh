cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r3_t13.log 2>&1 || { tail -40 gpurun_out/r3_t13.log; echo TESTS FAILED; exit 1; }
tail -3 gpurun_out/r3_t13.log
for wl in target c4 c3; do bash tools/ab.sh "--workload $wl" now nt now nt; done > gpurun_out/r3_ab13.log 2>&1; cat gpurun_out/r3_ab13.log
