"""Forward + backward as a HIP graph at several shapes, backward products on split rows (NW_BWD_SPLIT=1) against the
fp32 matrix cores (=0): where the default threshold of backward.hip:bwd_use_split belongs."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd import ops
dev = torch.device("cuda:0")


def graph_time(B, N, d, C, mode):
    os.environ["NW_BWD_SPLIT"] = mode
    g = torch.Generator().manual_seed(1)
    q = torch.randn(B, d, generator=g).to(dev).requires_grad_(True)
    s = torch.randn(N, d, generator=g).to(dev).requires_grad_(True)
    sy = (torch.arange(N) * C // N).to(dev)
    t = torch.randint(0, C, (B,), generator=g).to(dev)

    def step():
        q.grad = None; s.grad = None
        F.nll_loss(ops.nw_head(q, s, sy, C, "euclidean"), t).backward()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    q.grad = None; s.grad = None
    with torch.cuda.graph(graph):
        F.nll_loss(ops.nw_head(q, s, sy, C, "euclidean"), t).backward()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(30): graph.replay()
    torch.cuda.synchronize(); e0.record()
    for _ in range(200): graph.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 200 * 1e3


for B, N, d, C in [(16, 8192, 64, 10), (32, 512, 512, 10), (16, 256, 1024, 10), (1000, 256, 32, 5), (64, 1024, 64, 10), (64, 2048, 128, 10), (128, 2048, 128, 50), (128, 4096, 256, 100), (256, 4096, 128, 100),
                   (64, 10000, 512, 200), (256, 4096, 512, 200), (512, 4096, 256, 100), (256, 10000, 512, 200), (1024, 10000, 512, 200),
                   (256, 30000, 512, 200)]:
    t0, t1 = graph_time(B, N, d, C, "0"), graph_time(B, N, d, C, "1")
    print(f"B={B:5d} N={N:6d} d={d:4d}  B*N*d=2^{(B * N * d).bit_length() - 1}: fp32 cores {t0:7.1f} us, split rows {t1:7.1f} us", flush=True)
