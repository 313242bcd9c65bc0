import sys, torch
sys.path.insert(0, '.')
from nwhead_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(3)
for B, N, d, C in ((300, 5000, 64, 37), (64, 1000, 512, 200), (8, 20, 32, 5), (700, 3000, 36, 1000)):
    q, s = torch.randn(B, d, generator=g).to(dev), torch.randn(N, d, generator=g).to(dev)
    sy = torch.randint(0, C, (N,), generator=g).to(dev)
    a, wa = ops.nw_head(q, s, sy, C, return_weights=True)
    a, wa = a.clone(), wa.clone()
    for rep in range(3):
        b, wb = ops.nw_head(q, s, sy, C, return_weights=True)
        print(B, N, d, C, 'out equal', torch.equal(a, b), (a - b).abs().max().item(), 'w equal', torch.equal(wa, wb), (wa - wb).abs().max().item(),
              'ndiff', (a != b).sum().item(), (wa != wb).sum().item())
    sc1 = ops.nw_scores(q, s).clone()
    sc2 = ops.nw_scores(q, s)
    print('  scores equal', torch.equal(sc1, sc2))
