import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import T, load_golden
from procedural import fill_procedural
from nwhead_amd.model import load_model
import nwhead_amd.model.backbones as bb
g = load_golden("g6_backbones.npz")
dev = torch.device('cuda:0')
for name in ["resnet18", "CIFAR_ResNet18", "densenet121", "CIFAR_DenseNet121"]:
    x = T(g[f"{name}_x"])
    ref = g[f"{name}_train"]
    outs = {}
    for fused in (True, False):
        bb.FUSED_BN_RELU_TRAINING = fused
        net = fill_procedural(load_model(name)).to(dev).train()
        outs[fused] = net(x.to(dev).clone().requires_grad_(True)).detach().cpu().numpy()
    net = fill_procedural(load_model(name)).train()
    cpu = net(x.clone().requires_grad_(True)).detach().numpy()
    net64 = fill_procedural(load_model(name)).double().train()
    c64 = net64(x.double().clone().requires_grad_(True)).detach().numpy()
    sc = np.abs(ref).max()
    print(name, 'scale', sc, 'fused-vs-ref', np.abs(outs[True]-ref).max(), 'torchgpu-vs-ref', np.abs(outs[False]-ref).max(),
          'cpu-vs-ref', np.abs(cpu-ref).max(), 'fp64-vs-ref', np.abs(c64-ref).max(), 'fused-vs-fp64', np.abs(outs[True]-c64).max(),
          'torchgpu-vs-fp64', np.abs(outs[False]-c64).max())
