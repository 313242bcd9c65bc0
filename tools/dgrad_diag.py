import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd import ops
from nwhead_amd.model import load_model
from tests.procedural import fill_procedural_hash
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = load_model("resnet18"); fill_procedural_hash(net); net = net.to(dev).train()
convs = [(m.weight, m is not net.conv1) for m in net.modules() if isinstance(m, torch.nn.Conv2d)]
bank = ops.ConvWeightBank(convs); bank.refresh(force=True)
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
for name in ("layer4.1.conv2", "layer4.1.conv1", "layer4.0.conv2", "layer3.1.conv2", "layer1.1.conv2"):
    conv = dict(net.named_modules())[name]
    w = conv.weight
    cout, cin, k, _ = w.shape
    hw = {"layer4": 3, "layer3": 6, "layer1": 24}[name.split(".")[0]]
    for kind in ("random", "zero-mean"):
        x0 = cl(torch.randn(6, cin, hw, hw, device=dev))
        t = torch.randn(6, cout, hw, hw, device=dev)
        if kind == "zero-mean":
            t = t - t.mean((0, 2, 3), keepdim=True)
        t = cl(t)
        x64 = x0.double().requires_grad_(True)
        (F.conv2d(x64, w.double(), None, 1, 1) * t.double()).sum().backward()
        for b in (None, bank):
            x = x0.clone().requires_grad_(True)
            y = ops.conv2d_nhwc_train(x, w, 1, 1, operands=None if b is None else b.operands(w))
            (y * t).sum().backward()
            e = ((x.grad.double() - x64.grad).abs().max() / x64.grad.abs().max()).item()
            print(f"{name} {kind:9s} bank={b is not None}: dx err {e:.2e}  |w| max {float(w.abs().max()):.3e}")
