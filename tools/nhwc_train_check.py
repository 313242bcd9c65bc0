"""DenseNet training step: the channels-last path (own convolutions + BatchNorm kernels) against the NCHW path."""
import os, sys, time, torch, torch.nn.functional as F
sys.path.insert(0, ".")
import bench
from nwhead_amd import ops
from nwhead_amd.model import load_model
import nwhead_amd.model.backbones as BB
dev = torch.device("cuda:0")
arch = sys.argv[1] if len(sys.argv) > 1 else "densenet121"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
side = int(sys.argv[3]) if len(sys.argv) > 3 else 64
torch.manual_seed(0)
net = load_model(arch).to(dev).train()
x = torch.randn(n, 3, side, side, device=dev)
t = torch.randn(n, net.num_features, device=dev)

def run(flag):
    BB.NHWC_TRAINING = flag
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.reset_running_stats()
    net.zero_grad(set_to_none=True)
    out = net(x)
    (out * t).sum().backward()
    torch.cuda.synchronize()
    return out.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}, \
        {k: b.detach().clone() for k, b in net.named_buffers() if "running" in k}

o0, g0, b0 = run(False)
o1, g1, b1 = run(True)
print("absmax fallbacks:", ops._CONV_STATS["absmax_fallbacks"])
if os.environ.get("REF64") == "1":      # both paths against the same network in fp64 (NCHW, torch)
    import copy
    BB.NHWC_TRAINING = False
    net64 = copy.deepcopy(net).double()
    for m in net64.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.reset_running_stats()
    net64.zero_grad(set_to_none=True)
    BB.FUSED_BN_RELU_TRAINING = False
    o64 = net64(x.double())
    (o64 * t.double()).sum().backward()
    BB.FUSED_BN_RELU_TRAINING = True
    g64 = {k: p.grad for k, p in net64.named_parameters()}
    for name, o, g in (("NCHW", o0, g0), ("NHWC", o1, g1)):
        fe = ((o.double() - o64).abs().max() / o64.abs().max()).item()
        ge = max(((g[k].double() - g64[k]).abs().max() / (g64[k].abs().max() + 1e-300)).item() for k in g64)
        print(f"{name} vs fp64: features {fe:.2e}, worst gradient {ge:.2e}")

print("features rel err", ((o1 - o0).abs().max() / o0.abs().max()).item())
worst = max(((g1[k] - g0[k]).abs().max() / (g0[k].abs().max() + 1e-30)).item() for k in g0)
wk = max(g0, key=lambda k: ((g1[k] - g0[k]).abs().max() / (g0[k].abs().max() + 1e-30)).item())
print("worst grad rel err", worst, wk)
print("running stats rel err", max(((b1[k] - b0[k]).abs().max() / (b0[k].abs().max() + 1e-30)).item() for k in b0))
if len(sys.argv) > 4:
    for flag in (False, True):
        BB.NHWC_TRAINING = flag
        def step():
            net.zero_grad(set_to_none=True)
            (net(x) * t).sum().backward()
        tt = bench.time_kernel_events(step, 5, warmup=5)
        print("NHWC" if flag else "NCHW", f"{tt * 1e3:.2f} ms per fwd+bwd")
