#!/bin/bash
# The round-end check on the GPU box: all GPU tests, then the default bench line with its sub-records.
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/full_tests.log 2>&1
echo "exit $?" >> gpurun_out/full_tests.log
tail -5 gpurun_out/full_tests.log
grep -q "exit 0" gpurun_out/full_tests.log || exit 1
python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err || { tail -20 gpurun_out/bench_full.err; exit 1; }
python -c "
import json
d=json.loads(open('gpurun_out/bench_full.json').read().strip().splitlines()[-1])
print('target', d['value'], d['ms_per_step'], d['roofline'])
for k,v in d.get('extra',{}).items():
    print(k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ('value','ms_per_step','ms','unit')}, v.get('roofline',{}).get('frac'))
print('cpu', d.get('cpu_baseline'))
"
