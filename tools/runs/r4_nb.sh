#!/bin/bash
# per-kernel times of two library builds (variants/lib_old.so, variants/lib_new.so) on one box
cd /tmp && export TMPDIR=/tmp
for v in old new old new; do
  GM_LIB_PATH=/root/repo/variants/lib_$v.so rocprofv3 --kernel-trace -d /root/repo/gpurun_out/prof_nb_$v -o r -- python3 /root/repo/bench.py --workload target --steps 10 --warmup 2 --no-extra > /root/repo/gpurun_out/prof_nb_$v.log 2>&1
  echo "== $v"; python3 /root/repo/tools/db_stats.py /root/repo/gpurun_out/prof_nb_$v/r_results.db 12 | grep -v "sys_\|hm_node\|copyBuffer\|fillBuffer"
done
