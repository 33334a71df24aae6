cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3_t10.log 2>&1 || { tail -40 gpurun_out/r3_t10.log; echo TESTS FAILED; exit 1; }
tail -4 gpurun_out/r3_t10.log
for wl in target c4 c2; do
python bench.py --workload $wl --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', round(d['value'],2), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['breakdown'].items()})"
done
