"""precompute-style inference of several backbones: the folded copy (own kernels) against the plain eval forward.  GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd.model import load_model, fold_batchnorm
dev = torch.device("cuda:0")
archs = sys.argv[1:] or ["resnet10", "resnet18", "resnet34", "resnet50", "densenet121", "densenet161", "densenet169", "densenet201",
                         "CIFAR_ResNet18", "CIFAR_DenseNet121"]
for arch in archs:
    torch.manual_seed(0)
    net = load_model(arch).to(dev).eval()
    side = 32 if arch.startswith("CIFAR") else 96
    x = torch.randn(6, 3, side, side, device=dev)
    with torch.no_grad():
        for m in net.modules():             # non-trivial running statistics
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.1)
        ref = net(x)
        for cl in ((False, True) if "resnet" in arch.lower() and not arch.startswith("CIFAR") else (False,)):   # (NWNet keeps only the ResNets' copy channels_last)
            f = fold_batchnorm(net)
            if cl:
                f = f.to(memory_format=torch.channels_last)
            out = f(x.contiguous(memory_format=torch.channels_last) if cl else x)
            print(f"{arch:20s} folded{' channels_last' if cl else '':14s}: rel err {float((out - ref).abs().max() / ref.abs().max()):.2e}", flush=True)
