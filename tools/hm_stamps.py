"""Development tool: phase timeline of hm_node_kernel<128, 1, 4> (four workgroups, wave 0) from a -DHM_STAMPS build.

    GM_HM_FLAGS="-DHM_STAMPS" python -m gnn_manip_amd.build --tag=hmstamps
    GM_LIB_PATH=variants/lib_hmstamps.so python tools/hm_stamps.py [n_particles]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_manip_amd import EncProcDecGNN, _lib, get_connectivity  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda:0")
m = EncProcDecGNN(25, 4, 3, int(os.environ.get("HM_STAMPS_HIDDEN", "128")), 2, int(os.environ.get("HM_STAMPS_STEPS", "2"))).to(dev)
rng = np.random.default_rng(0)
side = (n / 5000) ** (1 / 3) * 0.13
pos = torch.tensor(rng.uniform(0, side, (n, 3)), dtype=torch.float32, device=dev)
s, r = get_connectivity(pos, 0.015, 20)
ei = torch.stack((s, r))
x = torch.randn(n, 25, device=dev)
ea = torch.randn(int(ei.shape[1]), 4, device=dev)
with torch.no_grad():
    for _ in range(3):
        m.forward(x, ea, ei)
torch.cuda.synchronize()
L = _lib.lib()
if not hasattr(L, "gm_debug_hm_stamps"):
    raise SystemExit("this library was not built with -DHM_STAMPS")
L.gm_debug_hm_stamps.restype = C.c_int
buf = (C.c_ulonglong * (4 * 4 * 16))()
assert L.gm_debug_hm_stamps(buf) == 0
st = np.array(buf, dtype=np.int64).reshape(4, 4, 16)
EDGE = os.environ.get("HM_STAMPS_EDGE") == "1"   # the stamps of hm_edge_kernel (hidden 256: the processor edge kernel of C4)
names = ["tile start", "e rows -> image", "P gather issue", "GEMM 1 (+ P arrival)", "hidden Linears", "LN exchange", "epilogue (LN, residual, stores, scatter-add)",
         "", "", "", ""] if EDGE else ["tile start", "h rows -> image", "GEMM 1a (h)", "agg rows -> image", "GEMM 1b (agg)", "hidden Linears", "LN publish + barrier",
         "epilogue (residual, h stores)", "h -> image", "projection half 0 (+ P stores)", "projection half 1 (+ P stores)"]
# the last launch that wrote the stamps is the last node kernel with a projection tail (step 1 of 2 has the decoder tail instead)
t_min = min(int(st[w, 0, 0]) for w in range(4) if st[w, 0, 0] > 0)
for w in range(4):
    last = max(int(v) for v in st[w].ravel())
    print(f"workgroup {64 * w}: first stamp at +{int(st[w, 0, 0]) - t_min} counts, last stamp at +{last - t_min}")
for wg in range(4):
    for t in range(4):
        row = st[wg, t]
        if row[0] == 0 or row[1] <= row[0]:
            continue
        d = np.diff(row[:11])
        tot = (row[6] - row[0]) if EDGE else (row[10] - row[0] if row[10] > row[0] else row[8] - row[0])
        print(f"workgroup {64 * wg} tile {t}: total {tot} counts (shader-clock counts, ~2.35 GHz)")
        for k in range(6 if EDGE else 10):
            if row[k + 1] > row[k]:
                print(f"    {names[k + 1]:34s} {d[k]:7d}  {100.0 * d[k] / max(tot, 1):5.1f} %")
