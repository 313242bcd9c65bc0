"""ResNet-18 + NW head training step (32 queries + 10 supports @224, as K4's shape) on the channels-last path and on the NCHW / MIOpen
path beside it; a few steps, for timing or a kernel trace.  usage: python tools/resnet_train_step.py [steps] [arch] [size] [queries]"""
import os, sys, time, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd.model import load_model
from nwhead_amd.nwhead.kernel import get_kernel
from nwhead_amd.nwhead.nw import NWHead
import nwhead_amd.optim as O
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
arch = sys.argv[2] if len(sys.argv) > 2 else "resnet18"
size = int(sys.argv[3]) if len(sys.argv) > 3 else 224
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 32
net = load_model(arch).to(dev).train()
opt = O.SGD(net.parameters(), lr=0.01, momentum=0.9, nesterov=True, weight_decay=1e-4)
xq = torch.randn(nq, 3, size, size, generator=g).to(dev); yq = torch.randint(0, 10, (nq,), generator=g).to(dev)
xs = torch.randn(10, 3, size, size, generator=g).to(dev); ys = torch.arange(10).to(dev)
head = NWHead(get_kernel("euclidean"), 10)
def step():
    opt.zero_grad(set_to_none=True)
    feats = net(torch.cat((xq, xs)))
    loss = F.nll_loss(head(feats[:nq], feats[nq:], ys), yq)
    loss.backward()
    opt.step()
    return loss
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    l = step()
torch.cuda.synchronize()
print(f"{arch} step {(time.perf_counter() - t0) / steps * 1e3:.2f} ms wall, loss {float(l.detach()):.4f}")
