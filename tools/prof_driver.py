#!/usr/bin/env python3
"""Tiny driver for rocprofv3: runs the forward at one shape a few times.
usage: python3 tools/prof_driver.py B N d C scores|fwd|partial iters"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nwhead_amd import ops

B, N, d, C = (int(a) for a in sys.argv[1:5])
what, iters = sys.argv[5], int(sys.argv[6])
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
q = torch.randn(B, d, generator=g).to(dev)
s = torch.randn(N, d, generator=g).to(dev)
sy = (torch.arange(N) % C).sort().values.to(dev)
cache = ops.SplitBank(s)
for _ in range(iters):
    if what == "scores":
        ops.nw_scores(q, s)
    elif what == "partial":
        ops.nw_partials(q, s, sy, C)
    elif what == "fwd_nonorm":
        ops.nw_head(q, s, sy, C)
    else:
        ops.nw_head(q, s, sy, C, support_cache=cache)
torch.cuda.synchronize()
print("done", what, B, N, d, C, iters)
