cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r3_t7.log 2>&1 || { tail -60 gpurun_out/r3_t7.log; echo TESTS FAILED; exit 1; }
tail -14 gpurun_out/r3_t7.log
timeout -k 10 600 python bench.py > gpurun_out/r3_bench7.json 2> gpurun_out/r3_bench7.err || { tail -20 gpurun_out/r3_bench7.err; echo BENCH FAILED; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_bench7.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', {k:d['roofline'][k] for k in ('bound','frac','avg_launch_ms','launches_timed')})
print('breakdown', d['breakdown'])
for k,v in d['extra'].items(): print(k, round(v['value'],2), v.get('ms', v.get('ms_per_step')))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['host_logical_cpus'])
PY
