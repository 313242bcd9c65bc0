import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nwhead_amd import ops
dev = torch.device("cuda:0")
for B, N, d in ((256, 10000, 1000), (256, 10000, 100), (4096, 50000, 1000), (256, 10000, 1024)):
    q, s = torch.randn(B, d, device=dev), torch.randn(N, d, device=dev)
    sy = (torch.arange(N, device=dev) * 200 // N)
    bank = ops.SplitBank(s, labels=sy)
    t0 = bench.time_kernel_events(lambda: ops.nw_head(q, s, sy, 200), 10, warmup=3, min_warm_ms=5)
    t1 = bench.time_kernel_events(lambda: ops.nw_head(q, s, sy, 200, support_cache=bank), 10, warmup=3, min_warm_ms=5)
    print(f"B={B} N={N} d={d}: no bank {t0 * 1e6:.1f} us, bank (pad {bank.pad}) {t1 * 1e6:.1f} us")
