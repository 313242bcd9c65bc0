"""Host cost of one ops.nw_head call (wall clock per call over back-to-back asynchronous calls of a tiny problem,
where the device is idle most of the time) next to the device time of K2's head shape."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for B, N, d, C in ((8, 64, 128, 10), (64, 1000, 512, 200)):
    q = torch.randn(B, d, generator=g).to(dev); s = torch.randn(N, d, generator=g).to(dev)
    sy = (torch.arange(N) % C).sort().values.to(dev)
    bank = ops.SplitBank(s, sy)
    for name, f in (("plain", lambda: ops.nw_head(q, s, sy, C)), ("bank", lambda: ops.nw_head(q, s, sy, C, support_cache=bank))):
        for _ in range(200): f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3000): f()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{B}x{N}x{d} {name}: host {1e6 * (t1 - t0) / 3000:.2f} us per call issued, {1e6 * (t2 - t0) / 3000:.2f} us per call completed")
