"""Development tool: opcode histogram of the innermost-labelled loops of one kernel (hipcc -save-temps .s file).

    python tools/isa_hist.py file.s kernel_fragment start_label [end_label_or_count]
Prints opcode counts of the instructions from start_label up to the backward branch to it.
"""
import collections
import re
import sys

path, frag, lab = sys.argv[1], sys.argv[2], sys.argv[3]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^[\w.$]+:", l) and frag in l and not l.startswith(".L"))
end = next(i for i in range(start + 1, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
i0 = next(i for i, l in enumerate(body) if l.strip().startswith(lab + ":"))
i1 = max(i for i, l in enumerate(body) if re.search(r"s_cbranch\w*\s+" + re.escape(lab) + r"\b", l) or re.search(r"s_branch\s+" + re.escape(lab) + r"\b", l))
ops = collections.Counter()
for l in body[i0:i1 + 1]:
    t = l.split(";")[0].strip()
    if not t or t.startswith(".") or t.endswith(":"):
        continue
    ops[t.split()[0]] += 1
tot = sum(ops.values())
print(f"{lab}: {tot} instructions")
for k, v in ops.most_common():
    print(f"  {k:32s} {v}")
