"""Random shapes through nw_conv1x1_f32 (LDS-DMA kernel, its 4-byte-DMA form, K split, the generic kernel) against fp64
torch: slab prefixes, pre scale/shift/ReLU, bias, post ReLU."""
import os, sys, random
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd import ops
dev = torch.device("cuda:0")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    n = rng.choice([1, 2, 7, 16, 64])
    cin = rng.choice([3, 16, 20, 64, 100, 224, 512, 992])
    cout = rng.choice([12, 32, 48, 128, 256, 512])
    h, w = rng.choice([(7, 7), (14, 14), (28, 28), (56, 56), (5, 9), (8, 8), (1, 1), (3, 5)])
    extra = rng.choice([0, 0, 8, 33])
    g = torch.Generator().manual_seed(it)
    full = torch.randn(n, cin + extra, h, w, generator=g).to(dev)
    x = full[:, :cin]
    wgt = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(dev)
    b = torch.randn(cout, generator=g).to(dev) if rng.random() < 0.5 else None
    pre = rng.random() < 0.6
    pa = (torch.rand(cin, generator=g) + 0.5).to(dev) if pre else None
    pb = torch.randn(cin, generator=g).to(dev) if pre else None
    prelu, postrelu = pre and rng.random() < 0.7, rng.random() < 0.5
    xx = x.double()
    if pre:
        xx = xx * pa.double().view(1, -1, 1, 1) + pb.double().view(1, -1, 1, 1)
        if prelu: xx = F.relu(xx)
    ref = torch.einsum("oc,nchw->nohw", wgt.double(), xx)
    if b is not None: ref = ref + b.double().view(1, -1, 1, 1)
    if postrelu: ref = F.relu(ref)
    out = ops.conv1x1(x, ops.pad_rows16(wgt.t().contiguous()), b, pa, pb, prelu, postrelu)
    err = ((out.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-9)).item()
    bad += err >= 2e-5
    print(f"{it:3d} n={n:3d} {cin:4d}->{cout:<4d} {h:2d}x{w:<2d} slab+{extra:<2d} pre={int(pre)}{int(prelu)} bias={int(b is not None)} post={int(postrelu)}: {err:.1e}{'' if err < 2e-5 else '   <-- CHECK'}", flush=True)
print("bad", bad)
