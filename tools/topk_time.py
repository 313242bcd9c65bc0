"""nw_topk_f32 at a few shapes: us per call (HIP events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nwhead_amd import ops
dev = torch.device("cuda:0")
for B, N, k in ((256, 10000, 10), (256, 10000, 20), (256, 16384, 20), (256, 20000, 20), (256, 50000, 20), (64, 50000, 200)):
    sc = torch.randn(B, N, device=dev)
    t = bench.time_kernel_events(lambda: ops.nw_topk(sc, k), 20, warmup=5, min_warm_ms=5)
    print(f"B={B} N={N} k={k}: {t * 1e6:.1f} us")
q = torch.randn(256, 512, device=dev)
for N in (10000, 50000):
    s = torch.randn(N, 512, device=dev)
    bank = ops.SplitBank(s)
    a = ops.nw_scores(q, s); b = ops.nw_scores(q, s, support_cache=bank)
    t0 = bench.time_kernel_events(lambda: ops.nw_scores(q, s), 20, warmup=5, min_warm_ms=5)
    t1 = bench.time_kernel_events(lambda: ops.nw_scores(q, s, support_cache=bank), 20, warmup=5, min_warm_ms=5)
    print(f"scores 256 x {N} x 512: fp32 kernel {t0 * 1e6:.1f} us, with the bank {t1 * 1e6:.1f} us, max |diff| {(a - b).abs().max().item():.1e}")
