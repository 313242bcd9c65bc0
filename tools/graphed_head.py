"""torch.cuda.make_graphed_callables over the NW head (forward and backward graphs behind an ordinary autograd call):
the eager training step without its host cost."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd import ops
B, N, d, C = 256, 10000, 512, 200
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
q = torch.randn(B, d, generator=g).to(dev).requires_grad_(True)
s = torch.randn(N, d, generator=g).to(dev).requires_grad_(True)
sy = (torch.arange(N) * C // N).to(dev)
t = torch.randint(0, C, (B,), generator=g).to(dev)


class Head(torch.nn.Module):
    def forward(self, q, s):
        return ops.nw_head(q, s, sy, C, "euclidean")


head = Head()
out = head(q, s); F.nll_loss(out, t).backward()
gq, gs = q.grad.clone(), s.grad.clone()
graphed = torch.cuda.make_graphed_callables(head, (q.detach().clone().requires_grad_(True), s.detach().clone().requires_grad_(True)))


def step(fn):
    q.grad = None; s.grad = None
    F.nll_loss(fn(q, s), t).backward()


def timeit(fn, n=200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20): step(fn)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): step(fn)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


step(graphed)
print("graphed gradients equal eager:", torch.equal(q.grad, gq), torch.equal(s.grad, gs))
print(f"eager {timeit(head):.1f} us, make_graphed_callables {timeit(graphed):.1f} us per forward+backward")
