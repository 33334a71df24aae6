"""Development tool: per-tick timeline of the three roles of sys_edge_kernel (workgroup 0, waves jb = 0) from a -DHEDGE_STAMPS build.

    GM_HEDGE_FLAGS="-DHEDGE_STAMPS" python -m gnn_manip_amd.build --tag=stamps
    GM_LIB_PATH=variants/lib_stamps.so python tools/sys_stamps.py [n_particles]

Ticks are s_memtime counts (about one per shader clock; the stamps themselves drain the scalar-memory counter, so the
figures are inflated).  DESIGN.md section 6 reads them.
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
buf = (C.c_ulonglong * (3 * 256 + 32))()
assert L.gm_debug_sys_stamps(buf) == 0
st_all = np.array(buf, dtype=np.int64)[:768].reshape(3, 32, 8)
real = np.array(buf, dtype=np.int64)[768:]   # s_memrealtime (100 MHz) at the start of role 0's ticks 16 .. 47
if real[0] > 0 and real[31] > real[0]:
    cyc, sec = st_all[0, 31, 0] - st_all[0, 0, 0], (real[31] - real[0]) / 100e6
    print(f"31 ticks: {cyc} s_memtime counts in {sec * 1e6:.2f} us of s_memrealtime -> {cyc / sec / 1e9:.3f} G counts/s "
          f"({sec / 31 * 1e6:.3f} us per tick)")
NAMES = [
    ["wait for the P rows + sum -> staging tile", "requests (P rows, indices) + tile -> accumulators", "24 MFMAs (Linear 1)",
     "ReLU + split -> image X1", "-", "wait at the tick barrier"],
    ["merge statistics", "e rows -> image E (waits for them)", "requests + 24 MFMAs (Linear 2) + LayerNorm / e_out between them",
     "ReLU + split -> image X2", "residual requests", "wait at the tick barrier"],
    ["merge statistics + scan flags", "accumulator init (bias)", "24 MFMAs (Linear 3) + scatter-add scan between them",
     "partial statistics + Z writes", "destinations of the next block", "wait at the tick barrier"],
]
print(f"N = {n}, E = {e}")
t0 = st_all[:, :, 0]
for role in range(3):
    st = st_all[role]
    ticks = [t for t in range(31) if st[t, 6] > st[t, 0] > 0 and st[t + 1, 0] > 0]
    print(f"role {role}: {len(ticks)} ticks averaged")
    for a in range(6):
        d = [st[t, a + 1] - st[t, a] for t in ticks]
        print(f"  {NAMES[role][a]:64s} {np.mean(d):8.0f}")
    print(f"  {'tick (start to start)':64s} {np.mean([st[t + 1, 0] - st[t, 0] for t in ticks]):8.0f}")
# arrival order at the barrier: stamp 5 (arrival) of each role relative to role 2's, and release (stamp 6)
ticks = [t for t in range(31) if all(st_all[r, t, 6] > st_all[r, t, 0] > 0 for r in range(3))]
if ticks:
    arr = np.array([[st_all[r, t, 5] - st_all[2, t, 0] for r in range(3)] for t in ticks])
    rel = np.array([[st_all[r, t, 6] - st_all[2, t, 0] for r in range(3)] for t in ticks])
    print("barrier arrival after role 2's tick start (roles 0, 1, 2):", arr.mean(axis=0).round(0), " release:", rel.mean(axis=0).round(0))

# per tick: which role's jb = 0 wave arrives last at the barrier, and by how much it trails the first one
if ticks:
    arr = np.array([[st_all[r, t, 5] - st_all[0, t, 0] for r in range(3)] for t in ticks])
    dur = np.array([st_all[0, t + 1, 0] - st_all[0, t, 0] for t in ticks if t + 1 < 32 and st_all[0, t + 1, 0] > 0])
    last = arr.argmax(axis=1)
    print("last role per tick:", "".join(str(int(x)) for x in last))
    print("spread (last - first arrival) per tick: mean %.0f  max %.0f;  tick duration: mean %.0f  std %.0f  min %.0f  max %.0f" % (
        (arr.max(axis=1) - arr.min(axis=1)).mean(), (arr.max(axis=1) - arr.min(axis=1)).max(), dur.mean(), dur.std(), dur.min(), dur.max()))
    for r in range(3):
        a = arr[:, r]
        print(f"role {r}: arrival after tick start mean {a.mean():.0f} std {a.std():.0f} min {a.min():.0f} max {a.max():.0f}")

# role 0, sub-phase of the requests: stamp 7 = the eight P requests issued (after stamp 1)
st = st_all[0]
tk = [t for t in range(31) if st[t, 7] > st[t, 1] > 0]
if tk:
    print("role 0: P requests issued %.0f cycles after the staging tile; index requests + tile reads + stamp %.0f more" % (
        np.mean([st[t, 7] - st[t, 1] for t in tk]), np.mean([st[t, 2] - st[t, 7] for t in tk])))
