"""Where the host time of one eager forward+backward of the head at T goes (cProfile over 2000 steps)."""
import cProfile, pstats, os, sys, time
import torch
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
def step():
    q.grad = None; s.grad = None
    F.nll_loss(ops.nw_head(q, s, sy, C, "euclidean"), t).backward()
for _ in range(50): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2000): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host issue time per step {1e6 * (t1 - t0) / 2000:.1f} us; with the final sync {1e6 * (t2 - t0) / 2000:.1f} us")
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
