"""Folded inference forward of a backbone over a batch: ms per forward.  python tools/fold_time.py ARCH [batch side]
(run once with NW_OWN_CONV3X3=1 and once without for an A/B on the same box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nwhead_amd.model import load_model, fold_batchnorm
arch = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
side = int(sys.argv[3]) if len(sys.argv) > 3 else 224
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = load_model(arch).to(dev).eval()
f = fold_batchnorm(m)
x = torch.randn(n, 3, side, side, device=dev)
if os.environ.get("CL") == "1":   # the channels_last form NWNet.enable_bn_folding gives the all-MIOpen ResNets
    f = f.to(memory_format=torch.channels_last)
    x = x.contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    ref = m(x); got = f(x)
    err = ((got - ref).abs().max() / ref.abs().max()).item()
    t = bench.time_kernel_events(lambda: f(x), 10, warmup=3, min_warm_ms=50)
print(f"{arch} n={n} {side}x{side} NW_OWN_CONV3X3={os.environ.get('NW_OWN_CONV3X3', '1')} NW_RESNET_OWN_CONV3X3={os.environ.get('NW_RESNET_OWN_CONV3X3', '0')} CL={os.environ.get('CL', '0')}: folded forward {t * 1e3:.3f} ms, "
      f"max |folded - plain| / max |plain| = {err:.1e}", flush=True)
