#!/bin/bash
# HBM traffic and SQ counters of the training kernels (one --pmc pass each)
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/prof_trpmc; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- python3 /root/repo/tools/bench_train.py --no-cpu-baseline --steps 3 > $O/$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq -- python3 /root/repo/tools/bench_train.py --no-cpu-baseline --steps 3 > $O/sq.log 2>&1
python3 - <<'PY'
import csv, glob, collections
O='/root/repo/gpurun_out/prof_trpmc'
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('FETCH_SIZE','WRITE_SIZE','sq'):
    for f in glob.glob(f'{O}/{d}/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][-40:]
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if not any(s in k for s in ('train_fwd_kernel<128, 2>','train_bwd_kernel<128, 1>','wgrad_kernel','segment_sum')): continue
    line=f"{k:42s}"
    for c in ('FETCH_SIZE','WRITE_SIZE'):
        if c in v: line+=f" {c}={sum(v[c])/len(v[c])/1e3:8.1f} MB(KB-units/1e3)"
    if 'SQ_BUSY_CYCLES' in v:
        busy=(sum(v['SQ_VALU_MFMA_BUSY_CYCLES'])/1024)/(sum(v['SQ_BUSY_CYCLES'])/32)
        line+=f" mfma_busy={busy:.2f} wait_any={sum(v['SQ_WAIT_ANY'])/sum(v['SQ_WAVE_CYCLES']):.2f} active_valu={sum(v['SQ_ACTIVE_INST_VALU'])/sum(v['SQ_WAVE_CYCLES']):.2f}"
    print(line)
PY
