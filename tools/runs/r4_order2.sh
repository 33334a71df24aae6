#!/bin/bash
# A/B on one box: renumbering off / on x workgroup numbering plain / XCD-major
for rep in 1 2; do
  for v in x0 x1; do for r in 0 1; do
    GM_RENUMBER=$r GM_LIB_PATH=variants/lib_$v.so python bench.py --workload target --steps 20 --warmup 5 --no-extra 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v renumber=$r', round(d['value'],2), round(d['ms_per_step'],3), d['breakdown'])
"
  done; done
done
