"""Forward against a prepared bank over a grid of shapes / class counts / label orders: us per call and TFLOP/s, to spot
performance cliffs (a path that is right but slow for some combination)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nwhead_amd import ops
dev = torch.device("cuda:0")
d = 512
for B, N in ((256, 10000), (4096, 10000), (256, 50000), (4096, 50000)):
    g = torch.Generator().manual_seed(1)
    q, s = torch.randn(B, d, generator=g).to(dev), torch.randn(N, d, generator=g).to(dev)
    for C in (1, 2, 10, 200, 1000, 5000):
        for pat in ("sorted", "random"):
            sy = ((torch.arange(N) * C // N) if pat == "sorted" else torch.randint(0, C, (N,), generator=g)).to(dev)
            bank = ops.SplitBank(s, labels=sy)
            try:
                t = bench.time_kernel_events(lambda: ops.nw_head(q, s, sy, C, support_cache=bank), 10, warmup=3, min_warm_ms=5)
                print(f"B={B:5d} N={N:6d} C={C:5d} {pat:6s}: {t * 1e6:8.1f} us  {2 * B * N * d / t / 1e12:6.1f} TFLOP/s", flush=True)
            except Exception as e:
                print(f"B={B:5d} N={N:6d} C={C:5d} {pat:6s}: {type(e).__name__}: {str(e)[:80]}", flush=True)
