import os, sys, cProfile, pstats, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, N, d, C = 64, 1000, 512, 200
q = torch.randn(B, d, generator=g).to(dev); s = torch.randn(N, d, generator=g).to(dev)
sy = (torch.arange(N) % C).sort().values.to(dev)
bank = ops.SplitBank(s, sy)
f = lambda: ops.nw_head(q, s, sy, C, support_cache=bank)
for _ in range(300): f()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3000): f()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
