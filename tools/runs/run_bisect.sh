#!/bin/bash
# Development: same-box bisect of a benchmark line over commits.  Build each commit in a worktree first:
#   for c in <shas>; do git worktree add -f variants/wt/$c $c; (cd variants/wt/$c && python -m gnn_manip_amd.build); done
# then  gpurun -- 'bash tools/runs/run_bisect.sh'  (variants/ is git-ignored but travels to the GPU box).
set -o pipefail
mkdir -p gpurun_out
R=$PWD
{
for c in adaf90f 595fbb7 b404c9e d27fcc7 HEAD; do
  if [ $c = HEAD ]; then d=$R; else d=$R/variants/wt/$c; fi
  cd $d
  for wl in c4; do
  python bench.py --workload $wl --steps 8 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c $wl', round(d['value'],2), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['breakdown'].items()})"
  done
done
} > $R/gpurun_out/bisect.log 2>&1
cat $R/gpurun_out/bisect.log
