"""gpurun_out/power/<workload>.txt (tools/micro/power_watch.sh: one rocm-smi sample per second while bench.py ran a long timed
region) -> profiles/<tag>_power.json: mean shader clock / package power under load per workload + the digest of the kernel sources
the library was built from (bench.py reports the figures only while that digest still matches the tree).

    python tools/power_summary.py r06
"""
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from gnn_manip_amd.build import source_digest  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src = os.path.join(root, "gpurun_out", "power")
out = {}
cap = None
if os.path.exists(os.path.join(src, "cap.txt")):
    m = re.search(r"([0-9]+(?:\.[0-9]+)?)\s*$", open(os.path.join(src, "cap.txt")).read().strip().splitlines()[0]) if open(os.path.join(src, "cap.txt")).read().strip() else None
    cap = float(m.group(1)) if m else None
for wl in ("target", "c2", "c3", "c4", "c5"):
    f = os.path.join(src, wl + ".txt")
    if not os.path.exists(f):
        continue
    rows = []
    for ln in open(f):
        p, c = re.search(r"power W ([0-9.]+)", ln), re.search(r"sclk ([0-9]+) MHz", ln)
        if p and c:
            rows.append((float(p.group(1)), float(c.group(1))))
    if not rows:
        continue
    # under load = samples at >= 85 % of the run's highest power (start-up, scene set-up and the tail are idle)
    top = max(r[0] for r in rows)
    load = [r for r in rows if r[0] >= 0.85 * top]
    rec = {"samples_under_load": len(load), "package_power_w": round(sum(r[0] for r in load) / len(load), 1),
           "package_power_min_w": min(r[0] for r in load), "package_power_max_w": max(r[0] for r in load),
           "sclk_mhz": round(sum(r[1] for r in load) / len(load), 1), "cap_w": cap}
    b = os.path.join(src, wl + ".bench.txt")
    if os.path.exists(b):
        lines = [l for l in open(b) if l.startswith("{")]
        if lines:
            d = json.loads(lines[-1])
            rec["bench_value_during_the_watch"] = d.get("value")
            rec["bench_steps"] = d.get("steps")
    out[wl] = rec
out["source_digest"] = source_digest()
out["note"] = ("rocm-smi --showpower --showclocks once a second while bench.py ran a long timed region of the workload "
               "(tools/micro/power_watch.sh); means over the samples at >= 85 % of the run's highest power")
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_power.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
