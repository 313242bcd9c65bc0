"""K4 step: host issue time (no sync inside the loop) beside the GPU time: is the eager step bound by the host?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
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
xin = torch.cat((xq, xs))
marks = {}
def step(stamp=False):
    t = [time.perf_counter()]
    opt.zero_grad(set_to_none=True)
    feats = dn(xin); t.append(time.perf_counter())
    loss = F.nll_loss(head(feats[:32], feats[32:], ys), yq); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    return t
for _ in range(4): step()
torch.cuda.synchronize()
# 1: each step synchronised before the next is issued -> host issue time of ONE step with an idle queue
acc = [0.0] * 4; n = 6
for _ in range(n):
    torch.cuda.synchronize()
    t = step()
    for i in range(4): acc[i] += t[i + 1] - t[i]
print("host issue per step, queue idle: forward %.2f ms, head %.2f, backward %.2f, optimizer %.2f, total %.2f" %
      tuple([1e3 * a / n for a in acc] + [1e3 * sum(acc) / n]), flush=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); t0 = time.perf_counter()
for _ in range(8): step()
t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize()
print("back to back: host %.2f ms/step, GPU %.2f ms/step" % (1e3 * (t1 - t0) / 8, e0.elapsed_time(e1) / 8), flush=True)
