cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
bash tools/ab.sh "--workload target" base nt2 p213 p123 p223 s4 s5 base nt2 p213 p123 p223 s4 s5 > gpurun_out/r3_ab14.log 2>&1; cat gpurun_out/r3_ab14.log
