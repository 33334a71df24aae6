"""Development: gradient error of the training path against float64 over a run of seeds, next to PyTorch float32's
(which seeds have a ReLU sign flip between float32-accurate evaluations).  python tools/runs/grad_seeds.py HIDDEN NUM_LAYERS M_STEPS"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np, torch
from conftest import BOUNDS, CART, CTRL, MAT, STATS
from oracle import epd_oracle as orc
from oracle import torch_epd
from gnn_manip_amd import EncProcDecGNN, scene
dev = torch.device("cuda:0")
KW = dict(stats=STATS, bounds=BOUNDS, conn_r=0.015, cartesian_idx=CART, material_idx=MAT)
def t(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
hidden, nl, m_steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for seed in range(200, 212):
    dims = (25, 4, 3, hidden, nl, m_steps)
    params = orc.init_params(*dims, seed)
    m = EncProcDecGNN(*dims); m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}); m = m.to(dev)
    obs = scene.make_scene(300, seed=seed, side=0.06)
    nodes, ea, s, r, _ = orc.process(obs, None, control_idx=CTRL, **KW)
    ei = np.stack((s, r))
    rng = np.random.default_rng(seed)
    target = rng.standard_normal((nodes.shape[0], 3)).astype(np.float32)
    out = m.forward(t(nodes), t(ea), t(ei))
    loss = torch.nn.functional.l1_loss(out, t(target), reduction="sum") / out.shape[0]
    loss.backward()
    ref_out, ref_loss, ref_g = torch_epd.loss_and_grads(params, nodes, ea, ei, target, nl, m_steps)
    _, _, g32 = torch_epd.loss_and_grads(params, nodes, ea, ei, target, nl, m_steps, torch.float32)
    errs, e32 = [], []
    for name, p in m.named_parameters():
        g, rr = p.grad.cpu().numpy(), ref_g[name]
        sc = max(np.abs(rr).max(), 1e-12)
        errs.append(np.abs(g - rr).max() / sc); e32.append(np.abs(g32[name] - rr).max() / sc)
    print(f"seed {seed}: E={ei.shape[1]} ours max {max(errs):.2e} med {np.median(errs):.2e} | torch-f32 max {max(e32):.2e} med {np.median(e32):.2e}")
