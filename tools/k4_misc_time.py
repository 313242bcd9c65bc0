"""Pieces of the K4 step that still run in torch: SGD (foreach vs fused), the stem's weight gradient by layout, the pools."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import bench
from nwhead_amd.model import load_model
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
dn = load_model("densenet121").to(dev).train()
for p in dn.parameters():
    p.grad = torch.randn_like(p) * 1e-3
for kw in ({}, {"fused": True}):
    try:
        opt = torch.optim.SGD(dn.parameters(), lr=0.01, momentum=0.9, nesterov=True, weight_decay=1e-4, **kw)
        t = bench.time_kernel_events(opt.step, 20, warmup=3)
        print("SGD", kw, f"{t*1e6:.1f} us", flush=True)
    except Exception as e:
        print("SGD", kw, "failed:", repr(e)[:200], flush=True)
x = torch.randn(42, 3, 224, 224, generator=g).to(dev)
gy_nhwc = torch.randn(42, 112, 112, 64, generator=g).to(dev).permute(0, 3, 1, 2)     # channels-last strides
def wg(gy, xx):
    return torch.ops.aten.convolution_backward(gy, xx, torch.empty(64, 3, 7, 7, device=dev), None, [2, 2], [3, 3], [1, 1], False,
                                               [0, 0], 1, [False, True, False])[1]
ref = wg(gy_nhwc.contiguous(), x)
for name, f in (("gy.contiguous(), x", lambda: wg(gy_nhwc.contiguous(), x)),
                ("gy channels_last view, x", lambda: wg(gy_nhwc, x)),
                ("gy channels_last, x channels_last", lambda: wg(gy_nhwc, x.contiguous(memory_format=torch.channels_last)))):
    try:
        out = f()
        err = float((out - ref).abs().max() / ref.abs().max())
        t = bench.time_kernel_events(f, 10, warmup=3)
        print(f"stem wgrad [{name}]: {t*1e6:.1f} us, rel diff {err:.2e}", flush=True)
    except Exception as e:
        print(f"stem wgrad [{name}] failed:", repr(e)[:200], flush=True)
# pools, channels-last
y = torch.randn(42, 112, 112, 64, generator=g).to(dev).permute(0, 3, 1, 2).requires_grad_(True)
o = F.max_pool2d(y, 3, 2, 1); go = torch.randn_like(o)
print("maxpool fwd %.1f us" % (1e6 * bench.time_kernel_events(lambda: F.max_pool2d(y, 3, 2, 1), 10, warmup=3)))
print("maxpool bwd %.1f us" % (1e6 * bench.time_kernel_events(lambda: torch.autograd.grad(o, y, go, retain_graph=True), 10, warmup=3)))
for (hw, c) in ((56, 128), (28, 256), (14, 512)):
    y = torch.randn(42, hw, hw, c, generator=g).to(dev).permute(0, 3, 1, 2).requires_grad_(True)
    o = F.avg_pool2d(y, 2, 2); go = torch.randn_like(o)
    tf = bench.time_kernel_events(lambda: F.avg_pool2d(y, 2, 2), 10, warmup=3)
    tb = bench.time_kernel_events(lambda: torch.autograd.grad(o, y, go, retain_graph=True), 10, warmup=3)
    print(f"avgpool {hw}x{hw}x{c}: fwd {tf*1e6:.1f} us, bwd {tb*1e6:.1f} us", flush=True)
