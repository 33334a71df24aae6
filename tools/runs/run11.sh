cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_domain.py -m gpu -x -q > gpurun_out/r3_t11.log 2>&1 || { tail -40 gpurun_out/r3_t11.log; echo TESTS FAILED; exit 1; }
tail -3 gpurun_out/r3_t11.log
bash tools/ab.sh "--workload target" pre now pre now > gpurun_out/r3_ab11.log 2>&1; cat gpurun_out/r3_ab11.log
bash tools/ab.sh "--workload c4" pre now pre now > gpurun_out/r3_ab11c4.log 2>&1; cat gpurun_out/r3_ab11c4.log
