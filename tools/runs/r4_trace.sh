#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_r4
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/target -- python3 $R/bench.py --workload target --no-cpu-baseline --no-extra > $O/target.log 2>&1
python3 - <<'PY'
import csv, glob, os
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/trace_r4"
f=max(glob.glob(f"{O}/target/*/*kernel_stats.csv"), key=os.path.getmtime)
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r:-float(r["TotalDurationNs"]))[:30]:
    print(f'{r["Name"][:90]:90s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:9.1f} us  total {float(r["TotalDurationNs"])/1e6:9.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}%')
PY
tail -1 $O/target.log | cut -c1-300
