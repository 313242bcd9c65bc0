"""Forward + backward of the head at T as a captured HIP graph (torch.cuda.CUDAGraph) against the eager loop:
the eager step is bound by the host (python, autograd, ~15 launches), the graph by the kernels."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd import ops

B, N, d, C = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (256, 10000, 512, 200)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
q = torch.randn(B, d, generator=g).to(dev).requires_grad_(True)
s = torch.randn(N, d, generator=g).to(dev).requires_grad_(True)
sy = (torch.arange(N) * C // N).to(dev)
t = torch.randint(0, C, (B,), generator=g).to(dev)


def step():
    q.grad = None; s.grad = None
    loss = F.nll_loss(ops.nw_head(q, s, sy, C, "euclidean"), t)
    loss.backward()
    return loss


def timeit(fn, n=200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20): fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(side)
ref_gq, ref_gs = q.grad.clone(), s.grad.clone()
print(f"eager: {timeit(step):.1f} us per forward+backward", flush=True)
graph = torch.cuda.CUDAGraph()
q.grad = None; s.grad = None
with torch.cuda.graph(graph):
    loss = F.nll_loss(ops.nw_head(q, s, sy, C, "euclidean"), t)
    loss.backward()
graph.replay(); torch.cuda.synchronize()
print("graph gradients equal eager:", torch.equal(q.grad, ref_gq), torch.equal(s.grad, ref_gs))
print(f"graph: {timeit(graph.replay):.1f} us per forward+backward", flush=True)
