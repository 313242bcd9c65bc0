"""Parity and timing probe of ops.conv2d_nhwc against torch's conv2d in fp64 (GPU box)."""
import sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from nwhead_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)


def check(n, cin, h, w, cout, k, stride, pad, bias=True, res=True, relu=True, tol=3e-6):
    x = (torch.randn(n, cin, h, w, generator=g) * 1.7 + 0.3).to(dev).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)).to(dev)
    b = torch.randn(cout, generator=g).to(dev) if bias else None
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    r = torch.randn(n, cout, ho, wo, generator=g).to(dev).contiguous(memory_format=torch.channels_last) if res else None
    sw = ops.SplitConvWeight(wt)
    y = ops.conv2d_nhwc(x, sw, b, r, relu, stride, pad)
    torch.cuda.synchronize()
    ref = F.conv2d(x.double(), wt.double(), None if b is None else b.double(), stride, pad)
    if r is not None:
        ref = ref + r.double()
    if relu:
        ref = ref.relu()
    err = (y.double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
    am = float(y.nw_amax.max())
    ok = err < tol and abs(am - float(y.abs().max())) <= 1e-6 * am
    print(f"n={n} cin={cin} {h}x{w} cout={cout} k={k} s={stride} p={pad}: rel err {err:.2e} amax {am:.4g} {'ok' if ok else 'FAIL'}", flush=True)
    return ok


def timeit(n, cin, h, w, cout, k, stride, pad, iters=20):
    x = torch.randn(n, cin, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, k, k, generator=g) * 0.05).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    sw = ops.SplitConvWeight(wt)
    am = ops.absmax(x)
    for _ in range(5):
        ops.conv2d_nhwc(x, sw, b, None, True, stride, pad, amax=am)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv2d_nhwc(x, sw, b, None, True, stride, pad, amax=am)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    wcl = wt.contiguous(memory_format=torch.channels_last)
    if cin == 3:
        xp = ops.to_nhwc_pad(x)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters):
            ops.conv2d_nhwc(xp, sw, b, None, True, stride, pad)
        e1.record(); torch.cuda.synchronize()
        print(f"   (stem on the padded 4-channel input alone: {e0.elapsed_time(e1) / iters * 1e3:.1f} us)")
    for _ in range(5):
        F.relu(F.conv2d(x, wcl, b, stride, pad))
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        F.relu(F.conv2d(x, wcl, b, stride, pad))
    e1.record()
    torch.cuda.synchronize()
    t2 = e0.elapsed_time(e1) / iters * 1e-3
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    fl = 2.0 * n * ho * wo * cout * cin * k * k
    print(f"time n={n} cin={cin} {h}x{w} cout={cout} k={k} s={stride}: ours {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF | torch cl {t2*1e6:8.1f} us {fl/t2/1e12:6.1f} TF", flush=True)


if __name__ == "__main__":
    ok = True
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    if mode in ("all", "check"):
        # PATCH mode: 3x3 s1 at the three tile shapes; tiles crossing rows and images; ragged last tile
        ok &= check(2, 32, 8, 8, 32, 3, 1, 1, bias=False, res=False, relu=False)
        ok &= check(3, 64, 9, 7, 64, 3, 1, 1)
        ok &= check(2, 64, 56, 56, 64, 3, 1, 1)
        ok &= check(4, 128, 28, 28, 128, 3, 1, 1)
        ok &= check(5, 256, 14, 14, 256, 3, 1, 1)
        ok &= check(9, 512, 7, 7, 512, 3, 1, 1)
        ok &= check(3, 128, 56, 56, 32, 3, 1, 1, bias=False, res=False, relu=False)
        ok &= check(3, 128, 14, 14, 32, 3, 1, 1, bias=False, res=False, relu=False)
        # GATHER mode: 1x1, strided
        ok &= check(2, 64, 56, 56, 128, 1, 1, 0, res=False)
        ok &= check(3, 256, 28, 28, 128, 1, 1, 0, bias=False, res=False, relu=False)
        ok &= check(2, 64, 56, 56, 128, 3, 2, 1)
        ok &= check(2, 64, 56, 56, 128, 1, 2, 0, res=False, relu=False)
        ok &= check(3, 992, 7, 7, 128, 1, 1, 0, bias=False, res=False, relu=False)
        ok &= check(2, 96, 10, 12, 64, 5, 1, 2)
        ok &= check(2, 32, 30, 30, 96, 3, 1, 0)
        # ROWRUN mode: few input channels (the stems)
        ok &= check(3, 3, 64, 64, 64, 7, 2, 3, res=False)
        ok &= check(2, 3, 224, 224, 64, 7, 2, 3, res=False)
        ok &= check(5, 3, 32, 32, 64, 3, 1, 1, res=False)
        ok &= check(2, 1, 28, 28, 32, 5, 1, 2, res=False)
        print("ALL OK" if ok else "SOME FAILED", flush=True)
    if mode in ("all", "time"):
        timeit(64, 3, 224, 224, 64, 7, 2, 3)
        timeit(64, 64, 56, 56, 64, 3, 1, 1)
        timeit(64, 128, 28, 28, 128, 3, 1, 1)
        timeit(64, 256, 14, 14, 256, 3, 1, 1)
        timeit(64, 512, 7, 7, 512, 3, 1, 1)
        timeit(64, 64, 56, 56, 128, 3, 2, 1)
        timeit(64, 64, 56, 56, 128, 1, 2, 0)
        timeit(42, 256, 56, 56, 128, 1, 1, 0)
        timeit(42, 128, 56, 56, 32, 3, 1, 1)
        timeit(42, 512, 28, 28, 128, 1, 1, 0)
        timeit(42, 128, 28, 28, 32, 3, 1, 1)
        timeit(42, 1024, 14, 14, 128, 1, 1, 0)
    if mode == "small":      # the latency-bound layers of DenseNet-121's last blocks (forward, and the data-gradient shapes)
        timeit(42, 128, 14, 14, 32, 3, 1, 1)
        timeit(42, 32, 14, 14, 128, 3, 1, 1)
        timeit(42, 512, 14, 14, 128, 1, 1, 0)
        timeit(42, 128, 14, 14, 512, 1, 1, 0)
        timeit(42, 128, 7, 7, 32, 3, 1, 1)
        timeit(42, 32, 7, 7, 128, 3, 1, 1)
        timeit(42, 800, 7, 7, 128, 1, 1, 0)
        timeit(42, 128, 7, 7, 800, 1, 1, 0)
        timeit(42, 32, 56, 56, 128, 3, 1, 1)
        timeit(42, 128, 56, 56, 256, 1, 1, 0)
    sys.exit(0 if ok else 1)
