#!/usr/bin/env python3
"""Driver for rocprofv3 --kernel-trace --stats: the kernels beside the tile kernel -- backward of the head at the T
shape (coefficients, two matrix-core products, reductions), support_influence (stand-alone over a rotation of buffers
beyond the Infinity Cache, and fused with the forward), top-k selection, merge."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from nwhead_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, N, d, C = 256, 10000, 512, 200
q = torch.randn(B, d, generator=g).to(dev).requires_grad_(True)
s = torch.randn(N, d, generator=g).to(dev).requires_grad_(True)
sy = (torch.arange(N) % C).sort().values.to(dev)
t = torch.randint(0, C, (B,), generator=g).to(dev)
qy = torch.randint(0, C, (B,), generator=g).to(dev)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(iters):
    q.grad = s.grad = None
    F.nll_loss(ops.nw_head(q, s, sy, C), t).backward()
qd, sd = q.detach(), s.detach()
bank = ops.SplitBank(sd, sy)
sets = [(torch.softmax(torch.randn(B, N, generator=g), -1).to(dev), torch.softmax(torch.randn(B, C, generator=g), -1).to(dev))
        for _ in range(32)]
for k in range(iters * 4):
    w, p = sets[k % 32]
    ops.support_influence_idx(p, qy, w, sy)
for _ in range(iters):
    ops.nw_head_influence(qd, sd, sy, C, qy, support_cache=bank)
    out, w = ops.nw_head(qd, sd, sy, C, return_weights=True, support_cache=bank)
sc = ops.nw_scores(qd, sd)
for _ in range(iters):
    ops.nw_topk(sc, 10)
torch.cuda.synchronize()
print("done")
