"""Turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the committed summaries under profiles/."""
import collections, csv, glob, json, os, shutil, sys


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
for wl in ("c2", "target", "train", "plan"):
    if not glob.glob(f"{src}/{wl}/*/*kernel_stats.csv"):
        continue
    f = newest(f"{src}/{wl}/*/*kernel_stats.csv")
    shutil.copy(f, f"{dst}/{tag}_{wl}_kernel_stats.csv")
out = {}
for wl in ("c2", "target"):
    d = {}
    # the processor edge kernel = the "<2, 1>" edge kernel with the largest total time in this workload's trace
    stats = list(csv.DictReader(open(f"{dst}/{tag}_{wl}_kernel_stats.csv")))
    EDGE = max((r for r in stats if "gm::edge_kernel" in r["Name"] and "<2, 1>" in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))["Name"]
    EDGE = EDGE[EDGE.index("gm::") + 4:EDGE.index("(")]
    d["kernel"] = EDGE
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = newest(f"{src}/pmc_{c}_{wl}/*/*counter_collection.csv")
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and EDGE in r["Kernel_Name"]]
        d[c + "_KB_per_launch"] = sum(v) / len(v)
        d["launches_" + c] = len(v)
    d["traffic_bytes_per_launch"] = (2 * d["FETCH_SIZE_KB_per_launch"] + d["WRITE_SIZE_KB_per_launch"]) * 1024
    d["traffic_bytes_uncorrected"] = (d["FETCH_SIZE_KB_per_launch"] + d["WRITE_SIZE_KB_per_launch"]) * 1024
    out[wl] = d
out["note"] = ("processor edge kernel of each workload (see its `kernel`); rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE doubled per "
               "MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B for 16-B-per-lane loads; the kernel's row gathers are not the "
               "calibrated streaming pattern, so the corrected figure is an upper bound); Infinity-Cache hits are counted")
json.dump(out, open(f"{dst}/{tag}_traffic.json", "w"), indent=1)
f = newest(f"{src}/pmc_sq_target/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{dst}/{tag}_target_pmc_sq.csv", "w") as fo:
    fo.write("kernel,counter,dispatches,mean_per_dispatch\n")
    for k in agg:
        if "gm::edge_kernel" not in k and "gm::node_kernel" not in k:
            continue
        for c, v in sorted(agg[k].items()):
            fo.write(f"\"{k}\",{c},{len(v)},{sum(v)/len(v):.6g}\n")
print(json.dumps(out, indent=1)[:700])
for wl in ("c2", "target"):
    print(open(f"{dst}/{tag}_{wl}_kernel_stats.csv").read().split("\n")[1][:200])
    print([l for l in open(f"{src}/{wl}.log") if l.startswith("{")][-1][:100])
