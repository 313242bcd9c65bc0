import sys, torch, numpy as np
sys.path.insert(0, '.')
from nwhead_amd import ops
from oracle import nw_oracle as O
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
B, N, d, C = 64, 2048, 128, 16
scale_s = (10.0 ** torch.randint(-2, 3, (N, 1), generator=g).float())
s = (torch.randn(N, d, generator=g) * scale_s).to(dev)
qs = 10.0 ** torch.randint(-2, 3, (B, 1), generator=g).float()
q = (torch.randn(B, d, generator=g) * qs).to(dev)
sy = (torch.arange(N) % C).sort().values.to(dev)
cache = ops.SplitBank(s)
for kind in ("euclidean", "cosine", "dotproduct"):
    fast = ops.nw_head(q, s, sy, C, kind, support_cache=cache).cpu().double()
    ref = O.nw_head_f64(q.cpu(), s.cpu(), sy.cpu(), C, kind)
    err = (fast - ref).abs()
    print(kind, 'max err', err.max().item(), 'rows with err>1e-4:', (err.max(1).values > 1e-4).nonzero().flatten().tolist()[:20])
    print('   q scales of bad rows', qs.flatten()[(err.max(1).values > 1e-4)].tolist()[:20])
