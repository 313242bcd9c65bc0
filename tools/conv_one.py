"""One convolution shape through ops.conv2d_nhwc, a few dozen calls (for rocprofv3): python tools/conv_one.py n cin h w cout k [stride pad]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd import ops
n, cin, h, w, cout, k = (int(a) for a in sys.argv[1:7])
stride = int(sys.argv[7]) if len(sys.argv) > 7 else 1
pad = int(sys.argv[8]) if len(sys.argv) > 8 else k // 2
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(n, cin, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
wt = (torch.randn(cout, cin, k, k, generator=g) * 0.05).to(dev)
sw = ops.SplitConvWeight(wt)
am = ops.absmax(x)
for _ in range(40):
    y = ops.conv2d_nhwc(x, sw, None, None, False, stride, pad, amax=am)
torch.cuda.synchronize()
print("done", tuple(y.shape))
