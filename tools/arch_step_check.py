"""One training step (forward + backward) of several backbones on the device paths in use, against the same step with every
own-kernel path off (torch / MIOpen): relative error of features and of the gradient vector.  GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nwhead_amd.model.backbones as BB
from nwhead_amd.model import load_model
dev = torch.device("cuda:0")
ARCHS = [a for a in sys.argv[1:]]
for arch, n, side in [(a, 4, 96) for a in ARCHS] or (("densenet169", 6, 96), ("densenet201", 4, 96), ("densenet121", 3, 64), ("resnet18", 8, 96), ("resnet34", 4, 96),
                      ("CIFAR_DenseNet121", 8, 32), ("CIFAR_ResNet18", 8, 32)):
    torch.manual_seed(0)
    try:
        net = load_model(arch).to(dev).train()
    except Exception as e:                      # an architecture name this build does not know
        print(arch, "skipped:", type(e).__name__, e)
        continue
    x = torch.randn(n, 3, side, side, device=dev)
    with torch.no_grad():
        t = torch.randn_like(net.eval()(x))
    net.train()
    res = []
    for nhwc, fused in ((True, True), (False, False)):
        BB.NHWC_TRAINING, BB.FUSED_BN_RELU_TRAINING = nhwc, fused
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
        net.zero_grad(set_to_none=True)
        out = net(x)
        (out * t).sum().backward()
        torch.cuda.synchronize()
        res.append((out.detach().double(), torch.cat([p.grad.detach().double().flatten() for p in net.parameters()])))
    BB.NHWC_TRAINING, BB.FUSED_BN_RELU_TRAINING = True, True
    (o1, g1), (o0, g0) = res
    cos = float((g1 * g0).sum() / (g1.norm() * g0.norm()))
    print(f"{arch:20s} features rel err {float((o1 - o0).abs().max() / o0.abs().max()):.2e}  gradient cosine {cos:.7f}", flush=True)
