# A/B of library variants (variants/lib_<tag>.so) on the training step: tools/ab_train.sh <rounds> tag1 tag2 ...   (interleaved rounds)
N=$1; shift
for r in $(seq 1 $N); do
  for t in "$@"; do
    GM_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_$t.so python tools/bench_train.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', round(d['value'],2), round(d['forward_ms'],3), round(d['backward_plus_optimizer_ms'],3))"
  done
done
