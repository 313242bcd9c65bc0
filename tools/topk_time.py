import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nwhead_amd import ops
dev = torch.device("cuda:0")
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for B, N in ((256, 50000), (64, 1000), (256, 10000)):
    q = torch.randn(B, 512, device=dev); s = torch.randn(N, 512, device=dev)
    sc = ops.nw_scores(q, s)
    print(f"B={B} N={N}: nw_topk k=20 {t(lambda: ops.nw_topk(sc, 20)):.1f} us | k=1024 {t(lambda: ops.nw_topk(sc, min(N, 1024))):.1f} us | "
          f"torch.argsort[:, :20] {t(lambda: torch.argsort(sc, dim=-1, descending=True, stable=True)[:, :20], 5):.1f} us | "
          f"torch.topk {t(lambda: torch.topk(sc, 20, dim=-1)):.1f} us | nw_scores {t(lambda: ops.nw_scores(q, s)):.1f} us")
