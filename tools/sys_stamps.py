"""Development tool: per-tick timeline of role 2 of sys_edge_kernel (workgroup 0, wave jb = 0) from a -DHEDGE_STAMPS build.

    GM_HEDGE_FLAGS="-DHEDGE_STAMPS" python -m gnn_manip_amd.build      # a stamps build of the library
    python tools/sys_stamps.py [n_particles]

Ticks are s_memtime counts (about one per shader clock).  Role 2 = Linear 3 + LayerNorm partial statistics + the
scatter-add's segmented scan; DESIGN.md section 6 explains why it is the long pole of a tick.
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
m = EncProcDecGNN(25, 4, 3, 128, 2, 2).to(dev)
rng = np.random.default_rng(0)
side = (n / 5000) ** (1 / 3) * 0.13
pos = torch.tensor(rng.uniform(0, side, (n, 3)), dtype=torch.float32, device=dev)
s, r = get_connectivity(pos, 0.015, 20)
ei = torch.stack((s, r))
e = int(ei.shape[1])
x = torch.randn(n, 25, device=dev)
ea = torch.randn(e, 4, device=dev)
with torch.no_grad():
    for _ in range(3):
        m.forward(x, ea, ei)
torch.cuda.synchronize()
L = _lib.lib()
if not hasattr(L, "gm_debug_sys_stamps"):
    raise SystemExit("this library was not built with -DHEDGE_STAMPS")
L.gm_debug_sys_stamps.restype = C.c_int
buf = (C.c_ulonglong * 256)()
assert L.gm_debug_sys_stamps(buf) == 0
st = np.array(buf, dtype=np.int64).reshape(32, 8)
ticks = [t for t in range(31) if st[t, 6] > st[t, 0] > 0 and st[t + 1, 0] > 0]
names = ["merge statistics + scan flags", "accumulator init (bias)", "24 MFMAs + scan between them", "partial statistics + Z writes",
         "index loads of the next block", "wait at the tick barrier"]
print(f"N = {n}, E = {e}; role 2, {len(ticks)} ticks averaged")
for a in range(6):
    d = [st[t, a + 1] - st[t, a] for t in ticks]
    print(f"  {names[a]:34s} {np.mean(d):8.0f}")
print(f"  {'tick (start to start)':34s} {np.mean([st[t + 1, 0] - st[t, 0] for t in ticks]):8.0f}")
