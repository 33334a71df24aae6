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
TILE = int(os.environ.get('GM_STAMP_TILE', '64'))
tiles = (n * 20 + TILE - 1) // TILE
buf = torch.zeros((tiles, 16), dtype=torch.int64, device=dev)
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
nt = (e + TILE - 1) // TILE
s = s[:nt]
t0 = s[:, 0].min()
names = ["layer1", "layers 2-3", "layernorm", "e_out stores", "c0 stage+barrier", "c0 segsum+barrier", "c0 stitch",
         "c1 stage+barrier", "c1 segsum+barrier", "c1 stitch", "tile end"]
ph = (s[:, 1:12] - s[:, 0:11]) / 100.0  # microseconds
print(f"tiles {nt}; kernel span {(s[:, 11].max() - t0) / 100.0:.1f} us")
for k, nm in enumerate(names):
    print(f"  {nm:18s} mean {ph[:, k].mean():7.2f} us   p10 {np.percentile(ph[:, k], 10):7.2f}  p90 {np.percentile(ph[:, k], 90):7.2f}")
print(f"  tile total         mean {((s[:, 11] - s[:, 0]) / 100.0).mean():7.2f} us")
# co-residency: group tiles by (xcc, hw_id cu/se/sh bits), look at overlap of MFMA phases
hw = (s[:, 14] >> 32) & 0xffff
xcc = s[:, 14] & 0xf
cu = (xcc << 16) | (hw & 0xff00)  # se_id, sh_id, cu_id bits
slot = hw & 0xf
print("wave slots seen:", np.unique(slot, return_counts=True))
key = cu[0]
sel = np.nonzero(cu == key)[0]
sel = sel[np.argsort(s[sel, 0])][:12]
print("timeline on one CU (us from kernel start): tile, block, slot, stamps 0..11")
for i in sel:
    print(i, s[i, 15], slot[i], np.round((s[i, 0:12] - t0) / 100.0, 1))
