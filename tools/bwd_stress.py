"""Random shapes through both backward paths (split-row products vs the fp32 matrix cores): max deviation relative to the
largest gradient entry, every kernel type.  A wider net than the test suite's fixed shapes."""
import os, sys, random
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd import ops
dev = torch.device("cuda:0")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
kinds = ["euclidean", "hypersphere_euclidean", "cosine", "dotproduct", "clip"]
worst = 0.0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    B = rng.choice([16, 17, 31, 64, 100, 129, 256, 300, 513, 1000])
    N = rng.choice([256, 257, 300, 1000, 1025, 4099, 10000, 20011, 36000])
    d = rng.choice([32, 64, 96, 160, 512, 1024])
    C = rng.choice([1, 2, 10, 200, 1000])
    if B * N * d < (1 << 22) or B * N * d > (1 << 33):
        continue
    kind = rng.choice(kinds)
    g = torch.Generator().manual_seed(it)
    q0, s0 = torch.randn(B, d, generator=g), torch.randn(N, d, generator=g)
    if kind == "dotproduct":
        q0, s0 = q0 * d ** -0.25, s0 * d ** -0.25
    if rng.random() < 0.3:
        s0 = s0 * torch.logspace(-2, 2, N).unsqueeze(1)
    sy = torch.randint(0, C, (N,), generator=g) if rng.random() < 0.5 else (torch.arange(N) * C // N)
    t = torch.randint(0, C, (B,), generator=g).to(dev)
    res = {}
    for mode in ("0", "1"):
        os.environ["NW_BWD_SPLIT"] = mode
        q, s = q0.to(dev).requires_grad_(True), s0.to(dev).requires_grad_(True)
        ls = torch.tensor(2.0, device=dev, requires_grad=True) if kind == "clip" else None
        F.nll_loss(ops.nw_head(q, s, sy.to(dev), C, kind, ls), t).backward()
        res[mode] = (q.grad, s.grad)
    dev_q = ((res["0"][0] - res["1"][0]).abs().max() / res["0"][0].abs().max().clamp_min(1e-20)).item()
    dev_s = ((res["0"][1] - res["1"][1]).abs().max() / res["0"][1].abs().max().clamp_min(1e-20)).item()
    fin = all(torch.isfinite(x).all().item() for p in res.values() for x in p)
    worst = max(worst, dev_q, dev_s)
    flag = "" if (dev_q < 1e-4 and dev_s < 1e-4 and fin) else "   <-- CHECK"
    print(f"{it:3d} {kind:12s} B={B:5d} N={N:6d} d={d:5d} C={C:5d}: gq {dev_q:.1e} gs {dev_s:.1e} finite {fin}{flag}", flush=True)
print("worst", worst)
