"""The stem's max pool (42 x 112 x 112 x 64, channels-last) forward / backward: own kernels beside torch's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import bench
from nwhead_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
y = torch.randn(42, 112, 112, 64, generator=g).to(dev).permute(0, 3, 1, 2).requires_grad_(True)
for name, f in (("own", ops.maxpool3s2_nhwc), ("torch", lambda t: F.max_pool2d(t, 3, 2, 1))):
    o = f(y); go = torch.randn(o.shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    tf = bench.time_kernel_events(lambda: f(y), 20, warmup=3)
    tb = bench.time_kernel_events(lambda: torch.autograd.grad(o, y, go, retain_graph=True), 20, warmup=3)
    print(f"max pool {name}: forward {tf*1e6:.1f} us, backward {tb*1e6:.1f} us", flush=True)
