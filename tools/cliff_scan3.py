"""Extreme aspect ratios of the forward (with and without a bank) and of support_influence / top-k."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nwhead_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for B, N, d, C in ((1, 50000, 512, 200), (8, 50000, 512, 200), (65536, 20, 512, 10), (65536, 26, 512, 10), (65536, 100, 128, 10),
                   (16384, 1000, 64, 1000), (2, 26, 2048, 2), (100000, 1000, 32, 5), (32, 200000, 128, 100), (1024, 200000, 512, 1000)):
    q, s = torch.randn(B, d, generator=g).to(dev), torch.randn(N, d, generator=g).to(dev)
    sy = (torch.arange(N) * C // N).to(dev)
    t0 = bench.time_kernel_events(lambda: ops.nw_head(q, s, sy, C), 5, warmup=2, min_warm_ms=2)
    bank = ops.SplitBank(s, labels=sy)
    t1 = bench.time_kernel_events(lambda: ops.nw_head(q, s, sy, C, support_cache=bank), 5, warmup=2, min_warm_ms=2)
    fl = 2 * B * N * d
    print(f"B={B:6d} N={N:6d} d={d:4d} C={C:4d}: no bank {t0 * 1e6:9.1f} us ({fl / t0 / 1e12:6.1f} TF/s)   bank {t1 * 1e6:9.1f} us ({fl / t1 / 1e12:6.1f} TF/s)", flush=True)
    del q, s, bank
for B, N, C in ((256, 10000, 200), (4096, 50000, 200), (16, 200000, 1000), (8192, 1000, 10)):
    w = torch.softmax(torch.randn(B, N, generator=g), -1).to(dev); p = torch.softmax(torch.randn(B, C, generator=g), -1).to(dev)
    qy = torch.randint(0, C, (B,), generator=g).to(dev); sy = (torch.arange(N) * C // N).to(dev)
    t = bench.time_kernel_events(lambda: ops.support_influence_idx(p, qy, w, sy), 10, warmup=2, min_warm_ms=2)
    print(f"influence B={B} N={N} C={C}: {t * 1e6:8.1f} us ({8 * B * N / t / 1e12:5.2f} TB/s)", flush=True)
