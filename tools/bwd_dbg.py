import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd import ops
dev = torch.device("cuda:0")
B, N, d, C = (int(x) for x in sys.argv[1:5])
sortedl = len(sys.argv) > 5 and sys.argv[5] == "sorted"
g = torch.Generator().manual_seed(28)
q0, s0 = torch.randn(B, d, generator=g), torch.randn(N, d, generator=g)
sy = (torch.arange(N) * C // N) if sortedl else torch.randint(0, C, (N,), generator=g)
t = torch.randint(0, C, (B,), generator=g)
q64, s64 = q0.double().to(dev).requires_grad_(True), s0.double().to(dev).requires_grad_(True)
sc = -torch.cdist(q64, s64)
p = torch.softmax(sc, -1) @ F.one_hot(sy.to(dev), C).double()
F.nll_loss(torch.log(p + 1e-12), t.to(dev)).backward()
for mode in ("0", "1"):
    os.environ["NW_BWD_SPLIT"] = mode
    q, s = q0.to(dev).requires_grad_(True), s0.to(dev).requires_grad_(True)
    F.nll_loss(ops.nw_head(q, s, sy.to(dev), C, "euclidean"), t.to(dev)).backward()
    eq = ((q.grad.double() - q64.grad).abs().max() / q64.grad.abs().max()).item()
    es_rows = (s.grad.double() - s64.grad).abs().amax(1) / s64.grad.abs().max()
    print(f"mode {mode}: gq err {eq:.2e}, gs err {es_rows.max().item():.2e}; worst rows {es_rows.topk(5).indices.tolist()} max|gs| {s64.grad.abs().max().item():.3e}")
