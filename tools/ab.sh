# A/B of library variants built into variants/lib_<tag>.so: tools/ab.sh "<bench args>" tag1 tag2 ...
ARGS=$1; shift
for t in "$@"; do
  GM_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_$t.so python bench.py $ARGS --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', round(d['value'],2), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['breakdown'].items() if k.endswith('ms_per_step')})"
done
