#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/variants
{
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_domain.py tests/test_gpu_fullsize.py tests/test_gpu_planner.py -x -q 2>&1 | tail -8
for rep in 1 2; do GM_LIB_PATH=$L/lib_noenc.so python bench.py --workload target --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('noenc', round(d['value'],2), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['breakdown'].items()})"
python bench.py --workload target --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('enc', round(d['value'],2), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['breakdown'].items()})"
done
python bench.py --workload c2 --steps 100 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('enc c2', round(d['value'],2), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['breakdown'].items()})"
} > gpurun_out/r4_t8.log 2>&1
cat gpurun_out/r4_t8.log
