"""K4 (DenseNet-121 joint forward + NW head + backward + SGD) in NCHW and channels_last."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import bench
from nwhead_amd.model import load_model
from nwhead_amd.nwhead.kernel import get_kernel
from nwhead_amd.nwhead.nw import NWHead
dev = torch.device("cuda:0")
from nwhead_amd.model import backbones
backbones.DENSE_INCREMENTAL_CAT = os.environ.get("NW_LIST_CAT", "0") != "1"
backbones.DENSE_PASSTHROUGH = os.environ.get("NW_NO_PASS", "0") != "1"
g = torch.Generator().manual_seed(7)
arch = sys.argv[1] if len(sys.argv) > 1 else "densenet121"
for fmt in (torch.contiguous_format,):
    dn = load_model(arch).to(dev).train().to(memory_format=fmt)
    opt = (torch.optim.SGD(dn.parameters(), lr=0.01, momentum=0.9, nesterov=True, weight_decay=1e-4, fused=True)
       if os.environ.get("NW_TORCH_SGD", "0") == "1" else
       __import__("nwhead_amd.optim", fromlist=["SGD"]).SGD(dn.parameters(), lr=0.01, momentum=0.9, nesterov=True, weight_decay=1e-4))
    xq = torch.randn(32, 3, 224, 224, generator=g).to(dev)
    yq = torch.randint(0, 10, (32,), generator=g).to(dev)
    xs = torch.randn(10, 3, 224, 224, generator=g).to(dev)
    ys = torch.arange(10).to(dev)
    head = NWHead(get_kernel("euclidean"), 10)
    def k4():
        opt.zero_grad(set_to_none=True)
        feats = dn(torch.cat((xq, xs)).contiguous(memory_format=fmt))
        loss = F.nll_loss(head(feats[:32], feats[32:], ys), yq)
        loss.backward()
        opt.step()
    ts = sorted(bench.time_kernel_events(k4, 5, warmup=3 if r == 0 else 0) for r in range(int(os.environ.get("NW_K4_REPEATS", "5"))))
    print(arch, fmt, f"{ts[0]*1e3:.2f} ms/step (best of {len(ts)} x 5 steps; median {ts[len(ts)//2]*1e3:.2f}, worst {ts[-1]*1e3:.2f})", flush=True)
    del dn, opt
