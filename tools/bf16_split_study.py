"""Numerical study (CPU, numpy): the whole encode-process-decode forward with every Linear evaluated as a sum of
bf16 x bf16 products of 3-way bf16 splits of both operands (fp32 accumulation), against float64.  Answers whether
the bf16 matrix pipe can carry the exact-fp32 MLPs: 9 and 6 products are as accurate as plain float32, 3 are not.

    python tools/bf16_split_study.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from conftest import BOUNDS, CART, CTRL, MAT, STATS
from oracle import epd_oracle as orc
from gnn_manip_amd import scene
KW = dict(stats=STATS, bounds=BOUNDS, conn_r=0.015, cartesian_idx=CART, material_idx=MAT)

def bf16(x):
    x = np.asarray(x, np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u >> 16) & 1) + 0x7FFF
    u = ((u + r) >> 16) << 16
    return (u.astype(np.uint32)).view(np.float32)

def split3(x):
    hi = bf16(x); r = (x - hi).astype(np.float32); mid = bf16(r); lo = bf16((r - mid).astype(np.float32))
    return [hi, mid, lo]

PAIRS = {9: [(i, j) for i in range(3) for j in range(3)], 6: [(0,0),(0,1),(1,0),(0,2),(2,0),(1,1)], 3: [(0,0),(0,1),(1,0)]}
MODE = None
def mm(x, w):  # x [rows,k] @ w.T [k,out]
    if MODE is None:
        return (x.astype(np.float32) @ w.T.astype(np.float32)).astype(np.float32)
    xs, ws = split3(x.astype(np.float32)), split3(w.astype(np.float32))
    out = np.zeros((x.shape[0], w.shape[0]), np.float32)
    # smallest terms first
    for i, j in sorted(PAIRS[MODE], key=lambda p: -(p[0] + p[1])):
        out += xs[i] @ ws[j].T
    return out

def mlp(p, prefix, x, nl, norm):
    for l in range(nl):
        x = np.maximum(mm(x, p[f"{prefix}.{2*l}.weight"]) + p[f"{prefix}.{2*l}.bias"], 0).astype(np.float32)
    k = 2 * nl
    x = (mm(x, p[f"{prefix}.{k}.weight"]) + p[f"{prefix}.{k}.bias"]).astype(np.float32)
    if norm:
        x = orc.layer_norm(x, p[f"{prefix}.{k+1}.weight"], p[f"{prefix}.{k+1}.bias"])
    return x

def fwd(p, nodes, ea, ei, nl, ms):
    j, i = ei[0], ei[1]
    h = mlp(p, "encoder.phi_node", nodes, nl, True); e = mlp(p, "encoder.phi_edge", ea, nl, True)
    for k in range(ms):
        en = mlp(p, f"processor.{k}.phi_edge", np.concatenate((h[i], h[j], e), 1), nl, True)
        agg = np.zeros_like(h); np.add.at(agg, i, en)
        hn = mlp(p, f"processor.{k}.phi_node", np.concatenate((h, agg), 1), nl, True)
        h, e = h + hn, e + en
    return mlp(p, "decoder", h, nl, False)

obs = scene.make_scene(400, seed=5, side=0.06)
params = orc.init_params(25, 4, 3, 128, 2, 10, 7)
nodes, ea, s, r, _ = orc.process(obs, None, control_idx=CTRL, **KW)
ei = np.stack((s, r))
import torch
from oracle import torch_epd
p64 = {k: torch.tensor(v, dtype=torch.float64) for k, v in params.items()}
ref = torch_epd.epd_forward(p64, torch.tensor(nodes, dtype=torch.float64), torch.tensor(ea, dtype=torch.float64), torch.tensor(ei), 2, 10).numpy()
for mode in (None, 9, 6, 3):
    MODE = mode
    out = fwd(params, nodes, ea, ei, 2, 10)
    print("mode", mode, "max rel err vs f64:", np.abs(out - ref).max() / np.abs(ref).max())
