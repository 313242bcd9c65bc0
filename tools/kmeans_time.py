"""Cluster stage of precompute() at the K3 bank: device k-means vs the sklearn call of the reference."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nwhead_amd.nwhead.utils import compute_clusters
N, d, C = 50000, 512, 200
g = torch.Generator().manual_seed(0)
x = torch.randn(N, d, generator=g) + 3 * torch.randn(C, d, generator=g).repeat_interleave(N // C, 0)
y = torch.arange(C).repeat_interleave(N // C)
xd, yd = x.cuda(), y.cuda()
for k in (1, 3):
    compute_clusters(xd, yd, k, backend="device"); torch.cuda.synchronize()
    t0 = time.perf_counter(); compute_clusters(xd, yd, k, backend="device"); torch.cuda.synchronize()
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter(); compute_clusters(x, y, k, backend="sklearn")
    t_sk = time.perf_counter() - t0
    print(f"k={k}: device {t_dev*1e3:8.1f} ms   sklearn (host, incl. D2H of the bank in real use) {t_sk*1e3:8.1f} ms", flush=True)
