"""Training-mode forward and backward time of the head at a few shapes (shared support)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from nwhead_amd import ops
import bench
dev = torch.device("cuda:0")
shapes = [(32, 10, 1024, 10), (32, 100, 1024, 10), (64, 1000, 512, 200), (256, 1000, 512, 200), (256, 10000, 512, 200),
          (1024, 4096, 512, 200)]
if len(sys.argv) > 4:
    shapes = [tuple(int(a) for a in sys.argv[1:5])]
for B, N, d, C in shapes:
    q, s, sy = bench.make_inputs(B, N, d, C, dev)
    q.requires_grad_(True); s.requires_grad_(True)
    t = torch.randint(0, C, (B,), device=dev)
    def fwd():
        return F.nll_loss(ops.nw_head(q, s, sy, C), t)
    def both():
        q.grad = s.grad = None
        fwd().backward()
    tf = bench.time_kernel_events(fwd, 20, warmup=5)
    tb = bench.time_kernel_events(both, 20, warmup=5)
    print(f"({B},{N},{d},{C}) fwd {tf*1e6:9.1f} us   fwd+bwd {tb*1e6:9.1f} us   bwd GEMM flops {4*B*N*d/1e9:7.2f} G -> "
          f"{4*B*N*d/max(tb-tf,1e-9)/1e12:6.2f} TFLOP/s", flush=True)
