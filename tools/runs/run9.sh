cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
python tools/runs/scan_seeds.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_planner.py tests/test_gpu_dataset.py -m gpu -q > gpurun_out/r3_t9.log 2>&1; tail -8 gpurun_out/r3_t9.log
python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('target', round(d['value'],2), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['breakdown'].items()})"
python bench.py --workload c4 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4', round(d['value'],2), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['breakdown'].items()})"
