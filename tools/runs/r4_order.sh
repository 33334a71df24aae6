#!/bin/bash
# A/B: particle numbering of the synthetic scene (random vs grid-cell order vs Morton order)
for rep in 1 2; do
  for o in "" linear morton; do
    GM_SCENE_ORDER=$o python bench.py --workload target --steps 20 --warmup 5 --no-extra 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('order=[$o]', round(d['value'],2), d['ms_per_step'], d['extra'].get('breakdown', d['extra']).get('edge_kernel_ms_per_step') if isinstance(d.get('extra'),dict) else '')
"
  done
done
