"""Development tool: write-only, read-only and copy rates of this GPU on buffers far larger than the Infinity Cache
(what the HBM system gives a kernel that streams 1 GB, to read the encoder's / edge kernel's times against)."""
import torch
dev = torch.device("cuda:0")
n = 256 * 1024 * 1024   # floats = 1 GiB
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev); c = torch.empty(2 * n, device=dev)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
gb = n * 4 / 1e9
w = t(lambda: a.fill_(1.0)); print(f"fill 1 GiB: {w:.3f} ms  {gb / w * 1e3:.0f} GB/s written")
r = t(lambda: a.sum()); print(f"sum  1 GiB: {r:.3f} ms  {gb / r * 1e3:.0f} GB/s read")
cp = t(lambda: b.copy_(a)); print(f"copy 1 GiB: {cp:.3f} ms  {2 * gb / cp * 1e3:.0f} GB/s read + written")
ad = t(lambda: torch.add(a, b, out=a)); print(f"a += b (2 reads, 1 write, in place): {ad:.3f} ms  {3 * gb / ad * 1e3:.0f} GB/s")
