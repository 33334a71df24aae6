#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
{
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_domain.py -x -q 2>&1 | tail -3
for wl in target c2 c4; do python bench.py --workload $wl --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', round(d['value'],2), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['breakdown'].items()})"; done
} > gpurun_out/r4_t12.log 2>&1
cat gpurun_out/r4_t12.log
