"""Numerical study (CPU, numpy): the whole encode-process-decode forward with every Linear evaluated as
lo*hi + hi*lo + hi*hi of two-way fp16 splits of both operands (fp32 accumulation), against float64 -- the arithmetic of
csrc/hedge.hip and csrc/hmlp.hip.  Variants: weights pre-scaled per Linear by a power of two (max|W| t in [0.25, 0.5), what
the kernels do) or not, fp16 subnormals kept (what v_mfma_f32_32x32x16_f16 does on gfx950) or flushed to zero.

    python tools/f16_split_study.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
from conftest import BOUNDS, CART, CTRL, MAT, STATS
from oracle import epd_oracle as orc
from oracle import torch_epd
from gnn_manip_amd import scene
KW = dict(stats=STATS, bounds=BOUNDS, conn_r=0.015, cartesian_idx=CART, material_idx=MAT)

FTZ = False
def f16(x):
    h = np.asarray(x, np.float32).astype(np.float16)
    if FTZ:
        h = np.where(np.abs(h.astype(np.float32)) < 6.1035e-5, np.float16(0), h)
    return h.astype(np.float32)

def split2(x):
    hi = f16(x)
    return hi, f16((x - hi).astype(np.float32))

MODE, SCALE = None, True
def mm(x, w):  # x [rows,k] @ w.T [k,out]
    x, w = x.astype(np.float32), w.astype(np.float32)
    if MODE is None:
        return (x @ w.T).astype(np.float32)
    t = np.float32(1.0)
    if SCALE:
        m = float(np.abs(w).max())
        if m > 0:
            t = np.float32(2.0 ** (-np.frexp(m)[1] - 1))
    xh, xl = split2(x)
    wh, wl = split2(w * t)
    out = (xl @ wh.T).astype(np.float32)
    out += xh @ wl.T
    out += xh @ wh.T
    return (out / t).astype(np.float32)

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
p64 = {k: torch.tensor(v, dtype=torch.float64) for k, v in params.items()}
ref = torch_epd.epd_forward(p64, torch.tensor(nodes, dtype=torch.float64), torch.tensor(ea, dtype=torch.float64), torch.tensor(ei), 2, 10).numpy()
for name, mode, scale, ftz in (("plain float32", None, False, False), ("fp16 x 3, scaled weights", 3, True, False),
                               ("fp16 x 3, unscaled weights", 3, False, False), ("fp16 x 3, scaled, subnormals flushed", 3, True, True)):
    MODE, SCALE, FTZ = mode, scale, ftz
    out = fwd(params, nodes, ea, ei, 2, 10)
    print(f"{name:40s} max rel err vs f64: {np.abs(out - ref).max() / np.abs(ref).max():.3e}")
