"""DenseNet-121 / ResNet-18 eval forward (precompute / predict side of the backbone): plain vs folded copy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nwhead_amd.model import load_model, fold_batchnorm
dev = torch.device("cuda:0")
arch = sys.argv[1] if len(sys.argv) > 1 else "densenet121"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
net = load_model(arch).to(dev).eval()
side = int(sys.argv[3]) if len(sys.argv) > 3 else 224
x = torch.randn(n, 3, side, side, device=dev)
folded = fold_batchnorm(net)
with torch.no_grad():
    ref = net(x)
    out = folded(x)
    print("max rel diff folded vs plain", float((out - ref).abs().max() / ref.abs().max()))
    t0 = bench.time_kernel_events(lambda: net(x), 10, warmup=3)
    t1 = bench.time_kernel_events(lambda: folded(x), 10, warmup=3)
print(f"{arch} eval fwd {n} images: plain {t0*1e3:.2f} ms   folded {t1*1e3:.2f} ms", flush=True)
