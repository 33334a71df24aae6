#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc2_r4
mkdir -p $O
B="--no-cpu-baseline --no-extra --workload target --steps 3 --warmup 1"
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum --output-format csv -d $O/ta -- python3 $R/bench.py $B > $O/ta.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum --output-format csv -d $O/tcp -- python3 $R/bench.py $B > $O/tcp.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TA_BUFFER_COALESCEABLE_WAVEFRONTS_sum TA_BUFFER_COALESCED_READ_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_WRITE_WAVEFRONTS_sum --output-format csv -d $O/ta2 -- python3 $R/bench.py $B > $O/ta2.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/pmc2_r4"
for d in ("ta","tcp","ta2"):
    fs=glob.glob(f"{O}/{d}/*/*counter_collection.csv")
    if not fs: print(d,"no file"); os.system(f"tail -3 {O}/{d}.log"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(max(fs,key=os.path.getmtime))):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in agg:
        if "sys_edge" in k or "hm_node_kernel<128, 1" in k or "sys_enc" in k:
            print(d,k[:40],{c:(len(v),round(sum(v)/len(v),1)) for c,v in agg[k].items()})
PY
