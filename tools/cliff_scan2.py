"""More cliff hunting: forward without a bank and forward + backward over embedding sizes, kernel types and class counts;
per-query (3-D) supports; support_influence; us per call and GFLOP/s of the products."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nwhead_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
print("-- forward (no bank) and forward+backward, B=256 N=10000")
for d in (32, 100, 128, 130, 512, 1000, 1024, 2048):
    for kind in ("euclidean", "cosine"):
        for C in (10, 1000, 20000):
            B, N = 256, 10000
            q, s = torch.randn(B, d, generator=g).to(dev), torch.randn(N, d, generator=g).to(dev)
            sy = (torch.arange(N) * C // N).to(dev)
            t = torch.randint(0, C, (B,), generator=g).to(dev)
            tf = bench.time_kernel_events(lambda: ops.nw_head(q, s, sy, C, kind), 5, warmup=2, min_warm_ms=2)
            qg, sg = q.clone().requires_grad_(True), s.clone().requires_grad_(True)
            def step():
                qg.grad = sg.grad = None
                F.nll_loss(ops.nw_head(qg, sg, sy, C, kind), t).backward()
            tb = bench.time_kernel_events(step, 5, warmup=2, min_warm_ms=2)
            print(f"d={d:5d} {kind:10s} C={C:6d}: fwd {tf * 1e6:8.1f} us ({2 * B * N * d / tf / 1e12:6.1f} TF/s)   fwd+bwd {tb * 1e6:8.1f} us", flush=True)
print("-- per-query supports (B, N, d), labels (B, N)")
for B, N, d, C in ((32, 40, 1024, 10), (256, 20, 512, 200), (64, 1000, 128, 10), (16, 5000, 512, 100)):
    q = torch.randn(B, d, generator=g).to(dev); s = torch.randn(B, N, d, generator=g).to(dev)
    sy = torch.randint(0, C, (B, N), generator=g).to(dev)
    tf = bench.time_kernel_events(lambda: ops.nw_head(q, s, sy, C), 5, warmup=2, min_warm_ms=2)
    print(f"B={B} N={N} d={d} C={C}: fwd {tf * 1e6:8.1f} us ({2 * B * N * d / tf / 1e9:8.1f} GF/s)", flush=True)
