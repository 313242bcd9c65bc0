"""Implicit-GEMM 3x3 convolution (nw_conv3x3_f32) against torch's conv2d (MIOpen) on the backbones' shapes at batch 64."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nwhead_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
shapes = [(64, 128, 32, 56), (64, 128, 32, 28), (64, 128, 32, 14), (64, 128, 32, 7), (64, 512, 32, 7), (64, 64, 64, 56), (64, 128, 128, 28), (64, 256, 256, 14), (64, 512, 512, 7)]
for n, cin, cout, side in shapes:
    x = torch.randn(n, cin, side, side, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).to(dev)
    wt = ops.conv3x3_weight(w)
    xcl = x.contiguous(memory_format=torch.channels_last); wcl = w.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        tf = bench.time_kernel_events(lambda: ops.conv3x3(x, wt, cin), 20, warmup=5, min_warm_ms=5)
        tc = bench.time_kernel_events(lambda: F.conv2d(x, w, padding=1), 20, warmup=8, min_warm_ms=5)
        tl = bench.time_kernel_events(lambda: F.conv2d(xcl, wcl, padding=1), 20, warmup=8, min_warm_ms=5)
    gf = 2 * cout * cin * 9 * side * side * n / 1e9
    print(f"n={n} {cin:4d}->{cout:<4d} {side:2d}x{side:<2d}: ours {tf*1e6:7.1f} us ({gf/tf/1e3:5.1f} TFLOP/s)   torch NCHW {tc*1e6:7.1f} us ({gf/tc/1e3:5.1f})   "
          f"torch channels_last {tl*1e6:7.1f} us ({gf/tl/1e3:5.1f})", flush=True)
