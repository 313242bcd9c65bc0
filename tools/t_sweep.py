"""Whole-forward time at T (B=256, N=10000, d=512, C=200) through ops.nw_head with a prepared bank: HIP events over
back-to-back calls.  usage: [NW_TILE_RS=n] [NW_SPLIT_QUERIES=1] python tools/t_sweep.py [B N d C]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd import ops
B, N, d, C = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (256, 10000, 512, 200)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
q = torch.randn(B, d, generator=g).to(dev); s = torch.randn(N, d, generator=g).to(dev)
sy = (torch.arange(N) % C).sort().values.to(dev)
bank = ops.SplitBank(s, sy)
f = lambda: ops.nw_head(q, s, sy, C, support_cache=bank)
for _ in range(300): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(5):
    e0.record()
    for _ in range(200): f()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
print(f"RS={os.environ.get('NW_TILE_RS','auto')} splitq={os.environ.get('NW_SPLIT_QUERIES','0')}  {B}x{N}x{d}: {best:.2f} us per forward")
