cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_domain.py -m gpu -x -q > gpurun_out/r3_t8a.log 2>&1; tail -25 gpurun_out/r3_t8a.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_train.py -m gpu -x -q > gpurun_out/r3_t8.log 2>&1 || { tail -40 gpurun_out/r3_t8.log; echo TESTS FAILED; exit 1; }
tail -4 gpurun_out/r3_t8.log
python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('target', round(d['value'],2), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['breakdown'].items()})"
python bench.py --workload c4 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4', round(d['value'],2), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['breakdown'].items()})"
