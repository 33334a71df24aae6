"""Diagnostic: per-tile phase stamps of the processor edge kernel (s_memrealtime, 100 MHz).
Prints phase durations and how the two workgroups sharing a CU are phased.  Not a timed run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gnn_manip_amd import EncProcDecGNN, GraphBoundedMultimaterialControl, RolloutEngine, _lib, scene

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda:0")
stats = dict(scene.STATS, acceleration_mean=[0.0, 0.0, 0.0])
obs = torch.from_numpy(scene.make_scene(n, seed=1000, vel_scale=1e-6)).to(dev)
torch.manual_seed(1234)
model = EncProcDecGNN(25, 4, 3, 128, 2, 10)
with torch.no_grad():
    model.decoder[-1].weight.mul_(1e-5); model.decoder[-1].bias.mul_(1e-5)
model = model.to(dev)
ga = GraphBoundedMultimaterialControl(scene.CONN_R, stats, scene.CART, scene.MAT, scene.CTRL, scene.BOUNDS)
eng = RolloutEngine(model, ga, n, device=dev)
eng.set_scene(obs)
L = _lib.lib()
tiles = (n * 20 + 127) // 128
buf = torch.zeros((tiles, 8), dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3):
        eng.step(obs, None)
    torch.cuda.synchronize()
    L.gm_debug_set_stamp_buffer(buf.data_ptr())
    eng.step(obs, None)
    torch.cuda.synchronize()
    L.gm_debug_set_stamp_buffer(None)
e = eng.status()
s = buf.cpu().numpy()
nt = (e + 127) // 128
s = s[:nt]
t0 = s[:, 0].min()
ph = (s[:, 1:6] - s[:, 0:5]) / 100.0  # microseconds
names = ["gather+layer1", "layers 2-3", "layernorm", "epilogue chunk0", "epilogue chunk1"]
print(f"tiles {nt}; kernel span {(s[:, 5].max() - t0) / 100.0:.1f} us")
for k, nm in enumerate(names):
    print(f"  {nm:18s} mean {ph[:, k].mean():7.2f} us   p10 {np.percentile(ph[:, k], 10):7.2f}  p90 {np.percentile(ph[:, k], 90):7.2f}")
print(f"  tile total         mean {((s[:, 5] - s[:, 0]) / 100.0).mean():7.2f} us")
# co-residency: group tiles by (xcc, hw_id cu/se/sh bits), look at overlap of MFMA phases
hw = (s[:, 6] >> 32) & 0xffff
xcc = s[:, 6] & 0xf
cu = (xcc << 16) | (hw & 0xff00)  # se_id, sh_id, cu_id bits
slot = hw & 0xf
print("wave slots seen:", np.unique(slot, return_counts=True))
key = cu[0]
sel = np.nonzero(cu == key)[0]
sel = sel[np.argsort(s[sel, 0])][:12]
print("timeline on one CU (us from kernel start): tile, block, slot, start, l1_end, l3_end, ln_end, c0_end, c1_end")
for i in sel:
    print(i, s[i, 7], slot[i], np.round((s[i, 0:6] - t0) / 100.0, 1))
