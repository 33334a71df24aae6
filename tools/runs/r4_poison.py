"""Diagnostic: the standalone InteractionNetwork forward after the caching allocator's free blocks were filled with a pattern
(NaN / huge / ones): a result that depends on memory nobody wrote in the call shows up as a mismatch against the oracle."""
import sys
import numpy as np
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import CTRL, CART, MAT, STATS, BOUNDS
from oracle import epd_oracle as orc
from gnn_manip_amd import EncProcDecGNN

dev = torch.device("cuda:0")
g4 = np.load("/root/repo/tests/golden/g4_features.npz")
KW = dict(stats=STATS, bounds=BOUNDS, conn_r=0.015, cartesian_idx=CART, material_idx=MAT)
nodes, ea, s, r, _ = orc.process(g4["obs_a"], None, control_idx=CTRL, **KW)
ei = np.stack((s, r))
params = orc.init_params(25, 4, 3, 128, 2, 10, 41)
m = EncProcDecGNN(25, 4, 3, 128, 2, 10)
m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
m = m.to(dev)
h0, e0 = orc.graph_independent(params, "encoder", nodes, ea, 2)
h1o, e1o = orc.interaction_network(params, "processor.0", h0, e0, ei, 2)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bad = 0
for it, pat in enumerate([float("nan"), 1e30, 1.0, -7.5, float("nan"), 3e38, 0.0, float("inf")] * 3):
    junk = [torch.full((n,), pat, device=dev) for n in (1 << 20, 1 << 22, 1 << 24, 3 << 18, 5 << 16, 7 << 12, 257, 65536 * 3)]
    ji = [torch.full((n,), 0x7fc00000 if it % 2 else 12345678, dtype=torch.int32, device=dev) for n in (1 << 20, 1 << 18, 4096, 1 << 22)]
    del junk, ji
    with torch.no_grad():
        h1, e1, _ = m.processor[0](t(h0), t(e0), t(ei))
    dh = np.abs(h1.cpu().numpy() - h1o).max(); de = np.abs(e1.cpu().numpy() - e1o).max()
    flag = "" if (dh < 1e-4 and de < 1e-4) else "   <-- MISMATCH"
    bad += bool(flag)
    print(f"iter {it} pattern {pat}: max |h1 - oracle| = {dh:.3g}, |e1 - oracle| = {de:.3g}{flag}")
print("mismatches:", bad)
