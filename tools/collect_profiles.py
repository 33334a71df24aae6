"""Turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the committed summaries under profiles/:
<tag>_<workload>_kernel_stats.csv (rocprofv3 --stats), <tag>_traffic.json (HBM bytes per launch of each workload's
processor edge kernel, read by bench.py), <tag>_<workload>_pmc_sq.csv (SQ counters of the model kernels)."""
import collections, csv, glob, json, os, shutil, sys


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


# rocprofv3 kernel-name fragment -> the name bench.py's roofline record uses
EDGE_KERNELS = {"sys_edge_kernel": "sys_edge_kernel", "hm_edge_kernel<64, false>": "hm_edge_kernel<64,false>",
                "hm_edge_kernel<128, false>": "hm_edge_kernel<128,false>", "hm_edge_kernel<256, false>": "hm_edge_kernel<256,false>"}
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
for wl in ("target", "c2", "c4", "train", "plan"):
    if not glob.glob(f"{src}/{wl}/*/*kernel_stats.csv"):
        continue
    shutil.copy(newest(f"{src}/{wl}/*/*kernel_stats.csv"), f"{dst}/{tag}_{wl}_kernel_stats.csv")
out = {}
for wl in ("target", "c2", "c4"):
    if not os.path.exists(f"{dst}/{tag}_{wl}_kernel_stats.csv"):
        continue
    stats = list(csv.DictReader(open(f"{dst}/{tag}_{wl}_kernel_stats.csv")))
    cands = [r for r in stats if any(k in r["Name"] for k in EDGE_KERNELS)]
    top = max(cands, key=lambda r: float(r["TotalDurationNs"]))
    frag = next(k for k in EDGE_KERNELS if k in top["Name"])
    same = [r for r in cands if frag in r["Name"]]   # every instantiation of the kernel (the last step's launch writes no e + e')
    calls = sum(int(r["Calls"]) for r in same)
    d = {"kernel": EDGE_KERNELS[frag], "stats_avg_ns": sum(float(r["TotalDurationNs"]) for r in same) / calls, "stats_calls": calls}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        if not glob.glob(f"{src}/pmc_{c}_{wl}/*/*counter_collection.csv"):
            continue
        f = newest(f"{src}/pmc_{c}_{wl}/*/*counter_collection.csv")
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and frag in r["Kernel_Name"]]
        d[c + "_KB_per_launch"] = sum(v) / len(v)
        d["launches_" + c] = len(v)
    if "FETCH_SIZE_KB_per_launch" in d and "WRITE_SIZE_KB_per_launch" in d:
        d["traffic_bytes_per_launch"] = (2 * d["FETCH_SIZE_KB_per_launch"] + d["WRITE_SIZE_KB_per_launch"]) * 1024
        d["traffic_bytes_uncorrected"] = (d["FETCH_SIZE_KB_per_launch"] + d["WRITE_SIZE_KB_per_launch"]) * 1024
    out[wl] = d
    if glob.glob(f"{src}/pmc_sq_{wl}/*/*counter_collection.csv"):
        f = newest(f"{src}/pmc_sq_{wl}/*/*counter_collection.csv")
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        with open(f"{dst}/{tag}_{wl}_pmc_sq.csv", "w") as fo:
            fo.write("kernel,counter,dispatches,mean_per_dispatch\n")
            for k in agg:
                if not any(s in k for s in ("sys_edge_kernel", "sys_node_kernel", "sys_proj_kernel", "hm_edge_kernel", "hm_node_kernel")):
                    continue
                for c, v in sorted(agg[k].items()):
                    fo.write(f"\"{k}\",{c},{len(v)},{sum(v)/len(v):.6g}\n")
                if "SQ_VALU_MFMA_BUSY_CYCLES" in agg[k] and "SQ_BUSY_CYCLES" in agg[k]:
                    # SQ_VALU_MFMA_BUSY_CYCLES sums over the 1024 SIMDs, SQ_BUSY_CYCLES over the 32 shader engines of the device
                    busy = (sum(agg[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024) / (sum(agg[k]["SQ_BUSY_CYCLES"]) / 32)
                    fo.write(f"\"{k}\",MFMA_BUSY_FRACTION = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (SQ_BUSY_CYCLES / 32 SEs),"
                             f"{len(agg[k]['SQ_BUSY_CYCLES'])},{busy:.4f}\n")
out["note"] = ("processor edge kernel of each workload (see its `kernel`); rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE "
               "doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B for 16-B-per-lane loads; the kernel's row gathers are not "
               "the calibrated streaming pattern, so the corrected figure is an upper bound); Infinity-Cache hits are counted")
sys.path.insert(0, root)
from gnn_manip_amd.build import source_digest  # noqa: E402
out["source_digest"] = source_digest()   # bench.py reports `traffic` only while the kernel sources still hash to this
json.dump(out, open(f"{dst}/{tag}_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
for wl in ("target", "c2", "c4"):
    if os.path.exists(f"{src}/{wl}.log"):
        print([l for l in open(f"{src}/{wl}.log") if l.startswith("{")][-1][:160])
