"""K2 (BASELINE config 2): ResNet-18 + NW head, predict over 64 images @224 against a 1000-row bank -- what NWNet.predict runs
after precompute() with enable_bn_folding(True): the folded channels_last copy (every convolution in nw_conv2d_nhwc_f16x2)
and the HIP head.  A few calls, for a kernel trace."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd import ops
from nwhead_amd.model import load_model, fold_batchnorm
from nwhead_amd.nwhead.kernel import get_kernel
from nwhead_amd.nwhead.nw import NWHead
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
net = load_model("resnet18").eval()
folded = fold_batchnorm(net).to(dev).to(memory_format=torch.channels_last)
x = torch.randn(64, 3, 224, 224, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
s = torch.randn(1000, 512, generator=g).to(dev)
sy = (torch.arange(1000) % 200).sort().values.to(dev)
head = NWHead(get_kernel("euclidean"), 200)
bank = ops.SplitBank(s, sy)
def k2():
    with torch.no_grad():
        return head(folded(x), s, sy, support_cache=bank)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(5):
    k2()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    out = k2()
torch.cuda.synchronize()
print(f"K2 predict: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per call, out {tuple(out.shape)}")
