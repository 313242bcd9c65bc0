"""Per-query supports (B, N, d) with labels (B, N): forward time and error against fp64, per kernel type (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nwhead_amd import ops
from oracle import nw_oracle as O
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for B, N, d, C in ((16, 5000, 512, 100), (64, 1000, 128, 10), (8, 20000, 512, 50), (256, 30, 512, 200)):
    q = torch.randn(B, d, generator=g); s = torch.randn(B, N, d, generator=g)
    sy = torch.randint(0, C, (B, N), generator=g)
    qd, sd, syd = q.to(dev), s.to(dev), sy.to(dev)
    for kind in ("euclidean", "cosine", "hypersphere_euclidean", "dotproduct"):
        out = ops.nw_head(qd, sd, syd, C, kind)
        ref = torch.stack([O.nw_head_f64(q[b:b + 1], s[b], sy[b], C, kind)[0] for b in range(min(B, 2))])
        err = (out[:min(B, 2)].cpu().double() - ref).abs().max().item()
        tf = bench.time_kernel_events(lambda: ops.nw_head(qd, sd, syd, C, kind), 10, warmup=3, min_warm_ms=2)
        ts = bench.time_kernel_events(lambda: ops.nw_scores(qd, sd, kind), 10, warmup=3, min_warm_ms=2)
        print(f"B={B} N={N} d={d} C={C} {kind:11s}: head {tf * 1e6:7.1f} us, scores alone {ts * 1e6:7.1f} us "
              f"({4.0 * B * N * d / ts / 1e12:5.2f} TB/s), max err vs fp64 {err:.2e}", flush=True)
