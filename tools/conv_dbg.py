import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from nwhead_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
n, cin, h, w, cout, k, s, p = [int(v) for v in sys.argv[1:9]]
x = (torch.randn(n, cin, h, w, generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
wt = (torch.randn(cout, cin, k, k, generator=g) * 0.1).to(dev)
sw = ops.SplitConvWeight(wt)
y = ops.conv2d_nhwc(x, sw, None, None, False, s, p)
torch.cuda.synchronize()
ref = F.conv2d(x.double(), wt.double(), None, s, p)
d = (y.double() - ref).abs().permute(0, 2, 3, 1).reshape(-1, cout)     # (pixels, cout)
bad = ~(d < 1e-3 * ref.abs().max())
rows = bad.any(1).nonzero().flatten()
print("pixels", d.shape[0], "bad pixels", rows.numel(), "nan", torch.isnan(y).sum().item())
if rows.numel():
    r = rows.tolist()
    tiles = sorted(set(v // 128 for v in r))
    print("bad tiles (128 px):", tiles[:40], "..." if len(tiles) > 40 else "")
    print("first bad rows:", r[:20], "bad channels of first:", bad[r[0]].nonzero().flatten().tolist()[:16])
