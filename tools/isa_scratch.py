"""Development tool: where a kernel's scratch (spill) traffic sits.  Per basic block of one kernel of an ISA listing (hipcc -S):
instruction count, scratch loads / stores, MFMAs, global loads / stores; blocks that are loop bodies are marked.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -Ignn_manip_amd/csrc gnn_manip_amd/csrc/hmlp.hip -o /tmp/hmlp.s
    python tools/isa_scratch.py /tmp/hmlp.s 'hm_edge_kernelILi256ELb0E'
"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + pat + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
lab, order, cnt = "entry", ["entry"], {"entry": dict(n=0, sl=0, ss=0, mfma=0, gl=0, gs=0, back=0)}
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        lab = m.group(1)
        order.append(lab)
        cnt[lab] = dict(n=0, sl=0, ss=0, mfma=0, gl=0, gs=0, back=0)
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    c = cnt[lab]
    c["n"] += 1
    c["sl"] += "scratch_load" in t
    c["ss"] += "scratch_store" in t
    c["mfma"] += "v_mfma" in t
    c["gl"] += "global_load" in t
    c["gs"] += "global_store" in t
    m = re.search(r"s_cbranch\S*\s+(\.LBB\d+_\d+)", t)
    if m and m.group(1) in cnt:   # a branch to a label already seen: loop back edge
        c["back"] = m.group(1)
for lab in order:
    c = cnt[lab]
    if c["n"] >= 8 or c["sl"] or c["ss"]:
        print(f"{lab:12s} n={c['n']:5d} scratch ld/st={c['sl']:3d}/{c['ss']:3d} mfma={c['mfma']:4d} global ld/st={c['gl']:3d}/{c['gs']:3d}" + (f"  loop-> {c['back']}" if c["back"] else ""))
