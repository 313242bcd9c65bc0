"""conv1 of a dense layer (1x1, c -> 128 over 42 x 14 x 14) three ways: nw_conv2d_nhwc_f16x2 on a dense t1; the same on a channel prefix
of a wide slab; nw_conv2d_nhwc_bnrelu_f16x2 (BatchNorm + ReLU in the loaders) on the prefix.  usage: python tools/conv_pre_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nwhead_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
f32 = dict(dtype=torch.float32, device=dev)
n, h, w, ctot, cout = 42, 14, 14, 1024, 128
rows = n * h * w
g = torch.Generator().manual_seed(0)
slab = torch.randn(rows, ctot, generator=g).to(dev)
st = ops._stream(slab)
for c in (256, 512, 992):
    wt = (torch.randn(cout, c, 1, 1, generator=g) / c ** 0.5).to(dev)
    sw = ops.SplitConvWeight(wt)
    t1 = slab[:, :c].contiguous()
    am_t = ops.absmax(t1)
    am_s = ops.absmax(slab)
    u = torch.empty(rows, cout, **f32)
    G = lib.nw_conv2d_nhwc_moments_groups(n, h, w, c, cout, 1, 1, 1, 0)
    part = torch.empty(5 * G * cout, **f32)
    tab = torch.cat([torch.zeros(c), torch.ones(c), torch.zeros(c)]).to(dev)
    def dense():
        lib.nw_conv2d_nhwc_f16x2(t1.data_ptr(), am_t.data_ptr(), sw.split.data_ptr(), sw.scale.data_ptr(), None, None, 0, u.data_ptr(), None,
                                 n, h, w, c, cout, 1, 1, 1, 0, 0, 0, part.data_ptr(), st)
    def prefix():
        lib.nw_conv2d_nhwc_f16x2(slab.data_ptr(), am_s.data_ptr(), sw.split.data_ptr(), sw.scale.data_ptr(), None, None, 0, u.data_ptr(), None,
                                 n, h, w, c, cout, 1, 1, 1, 0, ctot, 0, part.data_ptr(), st)
    def pre():
        lib.nw_conv2d_nhwc_bnrelu_f16x2(slab.data_ptr(), tab.data_ptr(), am_s.data_ptr(), 0, sw.split.data_ptr(), sw.scale.data_ptr(), None, 0,
                                        u.data_ptr(), None, n, h, w, c, cout, 1, 1, 1, 0, ctot, 0, part.data_ptr(), st)
    def nostat():
        lib.nw_conv2d_nhwc_bnrelu_f16x2(slab.data_ptr(), tab.data_ptr(), am_s.data_ptr(), 0, sw.split.data_ptr(), sw.scale.data_ptr(), None, 0,
                                        u.data_ptr(), None, n, h, w, c, cout, 1, 1, 1, 0, ctot, 0, None, st)
    ts = [bench.time_kernel_events(f, 50) * 1e6 for f in (dense, prefix, pre, nostat)]
    print(f"c={c}: dense t1 {ts[0]:.1f} us | slab prefix {ts[1]:.1f} | bnrelu on the prefix {ts[2]:.1f} | ... without moments {ts[3]:.1f}", flush=True)
