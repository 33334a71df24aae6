# ablation of the processor edge kernel on the target workload (GM_DEBUG_SKIP bits: 1 no MFMA layers,
# 2 no epilogue, 4 no aggregation, 8 no gathers; GM_DEBUG_LDS: extra dynamic LDS -> 1 workgroup / CU)
CFGS=${ABL_CFGS:-0:0 2:0 10:0 1:0 4:0 8:0}
for cfg in $CFGS; do
  s=${cfg%%:*}; l=${cfg##*:}
  GM_DEBUG_SKIP=$s GM_DEBUG_LDS=$l python bench.py --workload ${ABL_WL:-target} --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('skip=$s lds=$l', 'edge_ms', round(d['roofline']['avg_launch_ms'],4), 'step_ms', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],3), d['breakdown'])"
done
