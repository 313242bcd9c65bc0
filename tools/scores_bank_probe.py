import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd import ops
dev = torch.device("cuda:0")
q = torch.randn(256, 512, device=dev); s = torch.randn(10000, 512, device=dev)
bank = ops.SplitBank(s)
for _ in range(30):
    ops.nw_scores(q, s, support_cache=bank)
torch.cuda.synchronize()
