#!/bin/bash
# kernel-time budget of a rollout step at the target (current library)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_cur -o r -- python3 /root/repo/bench.py --workload target --steps 10 --warmup 2 --no-extra > /root/repo/gpurun_out/prof_cur.log 2>&1
python3 /root/repo/tools/db_stats.py /root/repo/gpurun_out/prof_cur/r_results.db 45 12
