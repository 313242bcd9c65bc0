"""DenseNet-121 eval-mode forward over 64 images @224 on the channels-last inference path (DenseNet._forward_nhwc_infer); a few
calls, for a kernel trace.  usage: python tools/dn_infer_step.py [calls]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd.model import load_model
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
net = load_model("densenet121").to(dev).eval()
x = torch.randn(64, 3, 224, 224, generator=g).to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
with torch.no_grad():
    for _ in range(3):
        net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        y = net(x)
    torch.cuda.synchronize()
print(f"DenseNet-121 eval forward: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per call, out {tuple(y.shape)}")
