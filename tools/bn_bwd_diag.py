import copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nwhead_amd.model.backbones as BB
from nwhead_amd import ops
from nwhead_amd.model import load_model
from tests.procedural import fill_procedural_hash
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = load_model("resnet18"); fill_procedural_hash(net); net = net.to(dev).train()
x = torch.randn(6, 3, 96, 96, device=dev)
t = torch.randn(6, 512, device=dev)
net64 = copy.deepcopy(net).double()
cap64 = {}
bn64 = net64.layer4[1].bn1
bn64.register_full_backward_hook(lambda m, gi, go: cap64.__setitem__("go", go[0].detach()))
BB.FUSED_BN_RELU_TRAINING = False; BB.RESNET_NHWC_TRAINING = False
(net64(x.double()) * t.double()).sum().backward()
BB.FUSED_BN_RELU_TRAINING = True; BB.RESNET_NHWC_TRAINING = True
caps = []
orig = ops._BNReLUNhwcFn.backward
def bw(ctx, gy, gpass=None):
    out = orig(ctx, gy, gpass)
    xv, wc, bc, mean, invstd = ctx.saved_tensors
    caps.append((xv.detach().clone(), gy.detach().clone(), wc, bc, mean, invstd, out[0].detach().clone(), out[1].detach().clone(), out[2].detach().clone(), ctx.relu))
    return out
ops._BNReLUNhwcFn.backward = staticmethod(bw)
(net(x) * t).sum().backward()
# backward order: the last BatchNorm first: layer4.1.bn2 (relu False), then layer4.1.bn1
for idx in (0, 1, 2):
    xv, gy, wc, bc, mean, invstd, dx, dg, db, relu = caps[idx]
    X, G = xv.double(), gy.double()
    m, i, a, b = mean.double().view(1, -1, 1, 1), invstd.double().view(1, -1, 1, 1), (wc.double() * invstd.double()).view(1, -1, 1, 1), bc.double().view(1, -1, 1, 1)
    pre = (X - m) * a + b
    gd = torch.where(pre > 0, G, torch.zeros_like(G)) if relu else G
    xh = (X - m) * i
    rows = X.shape[0] * X.shape[2] * X.shape[3]
    db64, dg64 = gd.sum((0, 2, 3)), (gd * xh).sum((0, 2, 3))
    dx64 = a * (gd - db64.view(1, -1, 1, 1) / rows - xh * dg64.view(1, -1, 1, 1) / rows)
    rel = lambda u, v: ((u.double() - v).norm() / v.norm()).item()
    print(f"bn #{idx} shape {tuple(xv.shape)} relu {relu}: kernel vs fp64-of-its-inputs: dx {rel(dx, dx64):.2e} dgamma {rel(dg, dg64):.2e} dbeta {rel(db, db64):.2e}")
    if idx == 1:
        print("   gy*mask vs the fp64 network's gradient at bn1's output:", rel(gd.float(), cap64["go"] * (cap64["go"] != 0)), " nnz", int((gd != 0).sum()), int((cap64["go"] != 0).sum()))
        mism = ((gd != 0) != (cap64["go"] != 0)).sum().item()
        print("   mask mismatches:", mism)
