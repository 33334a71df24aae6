cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc12
mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $O/counters.txt | sort -u | tr '\n' ' ' | head -c 6000; echo
B="--workload target --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py $B > $O/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $O/p$i.log)"
done
python3 - <<'PY'
import csv, glob, collections, os
O=os.environ.get('GRAFT_REPO_ROOT')+'/gpurun_out/pmc12'
for f in sorted(glob.glob(O+'/p*/*/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'sys_edge_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for c,v in sorted(agg.items()): print(c, len(v), '%.4g'%(sum(v)/len(v)))
PY
