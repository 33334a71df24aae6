"""Development tool: per-loop instruction census of one kernel in a hipcc -save-temps .s file.

    python tools/isa_loops.py file.s kernel_name_fragment

Splits the kernel's text at labels, finds the loop bodies (a label that a later branch jumps back to) and prints, per loop,
the count of instructions by class (MFMA, VALU, DPP, LDS read / write, global load / store, SALU, waitcnt ...) and the
s_waitcnt instructions in order.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "lds_write"
    if op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle"):
        return "lds_perm"
    if op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"):
        return "vmem_load"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("flat_store"):
        return "vmem_store"
    if op.startswith("global_atomic"):
        return "atomic"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime"):
        return "smem"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    path, frag = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^[\w.$]+:", l) and frag in l and not l.startswith(".L"))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    labels = {}
    insts = []
    for l in body:
        t = l.split(";")[0].strip()
        if not t or t.startswith("."):
            m = re.match(r"^(\.LBB[\w]+):", l.strip())
            if m:
                labels[m.group(1)] = len(insts)
            continue
        m = re.match(r"^(\.?[\w.$]+):", t)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        insts.append(t)
    # loops: backward branches
    loops = []
    for i, t in enumerate(insts):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", t)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i, tgt))
    print(f"kernel {frag}: {len(insts)} instructions, {len(loops)} backward branches")
    for a, b, tgt in loops:
        n = b - a + 1
        if n < 100:
            continue
        c = collections.Counter(classify(t.split()[0]) for t in insts[a:b + 1])
        dpp = sum(1 for t in insts[a:b + 1] if "_dpp" in t or "row_shr" in t or "row_bcast" in t)
        print(f"\nloop {tgt}: instructions {a}..{b} = {n}")
        print("  " + ", ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])) + f", (dpp {dpp})")
        if len(sys.argv) > 3 and sys.argv[3] == "-w":
            for t in insts[a:b + 1]:
                if t.startswith("s_waitcnt") or t.startswith("s_barrier"):
                    print("    ", t)


if __name__ == "__main__":
    main()
