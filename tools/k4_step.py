"""K4 (BASELINE config 4): DenseNet-121 + NW head training step, 32 queries + 10 supports @224; a few steps for a trace."""
import os, sys, time, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd.model import load_model
from nwhead_amd.nwhead.kernel import get_kernel
from nwhead_amd.nwhead.nw import NWHead
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
dn = load_model("densenet121").to(dev).train()
opt = (torch.optim.SGD(dn.parameters(), lr=0.01, momentum=0.9, nesterov=True, weight_decay=1e-4, fused=True)
       if os.environ.get("NW_TORCH_SGD", "0") == "1" else
       __import__("nwhead_amd.optim", fromlist=["SGD"]).SGD(dn.parameters(), lr=0.01, momentum=0.9, nesterov=True, weight_decay=1e-4))
xq = torch.randn(32, 3, 224, 224, generator=g).to(dev); yq = torch.randint(0, 10, (32,), generator=g).to(dev)
xs = torch.randn(10, 3, 224, 224, generator=g).to(dev); ys = torch.arange(10).to(dev)
head = NWHead(get_kernel("euclidean"), 10)
def k4():
    opt.zero_grad(set_to_none=True)
    feats = dn(torch.cat((xq, xs)))
    loss = F.nll_loss(head(feats[:32], feats[32:], ys), yq)
    loss.backward()
    opt.step()
    return loss
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for _ in range(3):
    k4()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    l = k4()
torch.cuda.synchronize()
print(f"K4 step {(time.perf_counter() - t0) / steps * 1e3:.2f} ms wall, loss {float(l):.4f}")
