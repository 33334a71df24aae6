import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import BOUNDS, CART, CTRL, MAT, STATS
from oracle import epd_oracle as orc
from gnn_manip_amd import EncProcDecGNN, scene
from gnn_manip_amd._lib import GMError
dev = torch.device("cuda:0")
KW = dict(stats=STATS, bounds=BOUNDS, conn_r=0.015, cartesian_idx=CART, material_idx=MAT)
def t(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
def model(p, dims, kernel):
    m = EncProcDecGNN(*dims); m.load_state_dict({k: torch.from_numpy(v) for k, v in p.items()}); m = m.to(dev); m.set_edge_kernel(kernel); return m
print("=== A: flag vs m_steps")
for n, side, seed in ((3000, 0.11, 61), (600, 0.07, 31)):
    obs = scene.make_scene(n, seed=seed, side=side)
    nodes, ea, s, r, _ = orc.process(obs, None, control_idx=CTRL, **KW)
    ei = np.stack((s, r))
    for ms in (1, 2, 3, 5, 10):
        p = orc.init_params(25, 4, 3, 128, 2, ms, seed)
        for kernel in ("sys", "hm"):
            m = model(p, (25, 4, 3, 128, 2, ms), kernel)
            with torch.no_grad():
                out = m.forward(t(nodes), t(ea), t(ei)).cpu().numpy()
            try:
                st = m.status()
            except GMError as e:
                st = "FLAG"
            ref = orc.epd_forward(p, nodes, ea, ei, 2, ms)
            print(n, ms, kernel, st, "nan" if np.isnan(out).any() else float(np.abs(out - ref).max() / np.abs(ref).max()))
print("=== B: encoder with features * 1e-6")
obs = scene.make_scene(600, seed=31, side=0.07)
nodes, ea, s, r, _ = orc.process(obs, None, control_idx=CTRL, **KW)
ei = np.stack((s, r))
for c in (1.0, 1e-6, 1e4):
    p = orc.init_params(25, 4, 3, 128, 2, 2, 528)
    c32 = np.float32(c)
    p["encoder.phi_node.0.weight"] = (p["encoder.phi_node.0.weight"] / c32).astype(np.float32)
    p["encoder.phi_edge.0.weight"] = (p["encoder.phi_edge.0.weight"] / c32).astype(np.float32)
    nc, ec = (nodes * c32).astype(np.float32), (ea * c32).astype(np.float32)
    m = model(p, (25, 4, 3, 128, 2, 2), "sys")
    h0, e0 = orc.graph_independent(p, "encoder", nc, ec, 2)
    with torch.no_grad():
        h, e, _ = m.encoder(t(nc), t(ec), t(ei))
    h, e = h.cpu().numpy(), e.cpu().numpy()
    print(c, "h err", np.abs(h - h0).max(), "e err", np.abs(e - e0).max(), "worst e row", int(np.abs(e - e0).max(axis=1).argmax()), ec[int(np.abs(e - e0).max(axis=1).argmax())])
