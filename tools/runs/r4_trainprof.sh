#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cat > /tmp/tp.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
import bench
r = bench.extra_train(torch.device("cuda:0"), steps=10, warmup=3)
print("train", round(r["value"], 2), "steps/s", round(r["ms"], 3), "ms")
PY
rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_tr -o r -- python3 /tmp/tp.py > /root/repo/gpurun_out/prof_tr.log 2>&1
tail -2 /root/repo/gpurun_out/prof_tr.log
python3 /root/repo/tools/db_stats.py /root/repo/gpurun_out/prof_tr/r_results.db 30 13
