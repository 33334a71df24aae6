#!/bin/bash
# PMC passes of the target workload (each counter set in its own run)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_r4
mkdir -p $O
B="--no-cpu-baseline --no-extra --workload target --steps 3 --warmup 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py $B > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py $B > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq -- python3 $R/bench.py $B > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -- python3 $R/bench.py $B > $O/tcc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq2 -- python3 $R/bench.py $B > $O/sq2.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/pmc_r4"
for d in ("fetch","write","sq","tcc","sq2"):
    fs=glob.glob(f"{O}/{d}/*/*counter_collection.csv")
    if not fs: print(d,"no file"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(max(fs,key=os.path.getmtime))):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in agg:
        if "sys_edge" in k or "hm_node_kernel" in k or "hm_edge" in k:
            print(d,k,{c:(len(v),round(sum(v)/len(v),1)) for c,v in agg[k].items()})
PY
