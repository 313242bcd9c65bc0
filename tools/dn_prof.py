"""DenseNet-121 folded inference forward over 64 images @224 (for rocprofv3 --kernel-trace): warm-up forwards (MIOpen's
first calls per configuration search and run stand-in kernels), a 200 ms pause, then REPS measured forwards; read the
trace with tools/trace_tail.py."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd.model import load_model, fold_batchnorm
dev = torch.device("cuda:0")
torch.manual_seed(0)
arch = sys.argv[2] if len(sys.argv) > 2 else "densenet121"
net = load_model(arch).to(dev).eval()
folded = fold_batchnorm(net)
x = torch.randn(64, 3, 224, 224, device=dev)
if len(sys.argv) > 3 and sys.argv[3] == "cl":   # what NWNet.enable_bn_folding(channels_last=True) runs for the ResNets
    folded = folded.to(memory_format=torch.channels_last)
    x = x.contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(12):
        folded(x)
    torch.cuda.synchronize()
    time.sleep(0.2)
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
        folded(x)
torch.cuda.synchronize()
print("done")
