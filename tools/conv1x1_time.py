"""Fused 1x1 convolution (nw_conv1x1_f32) against the unfused sequence it replaces (scale-shift-ReLU kernel, torch conv2d =
MIOpen/Tensile, bias, ReLU) on DenseNet-121's dense-layer shapes at batch 64.  usage: python tools/conv1x1_time.py"""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nwhead_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
shapes = [(64, 64, 56), (64, 224, 56), (64, 128, 28), (64, 480, 28), (64, 256, 14), (64, 992, 14), (64, 512, 7), (64, 992, 7)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
tot_f = tot_u = 0.0
for n, cin, side in shapes:
    cout = 128
    slab = torch.randn(n, cin + 32, side, side, generator=g).to(dev)
    x = slab[:, :cin]
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(dev)
    wt = ops.pad_rows16(w.t().contiguous())
    b = torch.randn(cout, generator=g).to(dev)
    a, s = (torch.rand(cin, generator=g) + 0.5).to(dev), torch.randn(cin, generator=g).to(dev)
    w4 = w[:, :, None, None].contiguous()
    with torch.no_grad():
        fused = lambda: ops.conv1x1(x, wt, b, a, s, pre_relu=True, post_relu=True)
        unfused = lambda: F.relu(F.conv2d(ops.scale_shift_relu(x, a, s), w4, b))
        conv_only = lambda: F.conv2d(x, w4)
        tf = bench.time_kernel_events(fused, 20, warmup=5, min_warm_ms=5)
        tu = bench.time_kernel_events(unfused, 20, warmup=5, min_warm_ms=5)
        tc = bench.time_kernel_events(conv_only, 20, warmup=5, min_warm_ms=5)
    gf = 2 * cout * cin * side * side * n / 1e9
    mb = (cin + cout) * side * side * n * 4 / 1e6
    tot_f += tf; tot_u += tu
    print(f"n={n} cin={cin:4d} {side:2d}x{side:<2d}: fused {tf*1e6:7.1f} us ({gf/tf/1e3:5.1f} TFLOP/s, {mb/tf/1e6:5.2f} TB/s)   "
          f"unfused {tu*1e6:7.1f} us   torch conv alone {tc*1e6:7.1f} us ({gf/tc/1e3:5.1f} TFLOP/s)", flush=True)
print(f"sum fused {tot_f*1e6:.0f} us   unfused {tot_u*1e6:.0f} us")
