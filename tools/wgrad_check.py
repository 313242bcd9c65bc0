"""Parity / timing probe of ops.conv2d_nhwc_wgrad against torch's convolution_backward in fp64."""
import sys, torch
sys.path.insert(0, ".")
from nwhead_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)

def ref(x, gy, wshape, pad):
    return torch.ops.aten.convolution_backward(gy.double().contiguous(), x.double().contiguous(), torch.empty(wshape, dtype=torch.float64, device=dev),
                                               None, [1, 1], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False])[1]

def check(n, cin, h, w, cout, k, tol=3e-6):
    x = (torch.randn(n, cin, h, w, generator=g) + 0.2).to(dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(n, cout, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    dw = ops.conv2d_nhwc_wgrad(x, gy, (cout, cin, k, k), 1, (k - 1) // 2)
    torch.cuda.synchronize()
    r = ref(x, gy, (cout, cin, k, k), (k - 1) // 2)
    err = ((dw.double() - r).abs().max() / r.abs().max()).item()
    print(f"wgrad n={n} cin={cin} {h}x{w} cout={cout} k={k}: rel err {err:.2e} {'ok' if err < tol else 'FAIL'}", flush=True)
    return err < tol

def timeit(n, cin, h, w, cout, k, iters=20):
    x = torch.randn(n, cin, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(n, cout, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    ax, ag = ops.absmax(x), ops.absmax(gy)
    for _ in range(3):
        ops.conv2d_nhwc_wgrad(x, gy, (cout, cin, k, k), 1, (k - 1) // 2, ax, ag)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv2d_nhwc_wgrad(x, gy, (cout, cin, k, k), 1, (k - 1) // 2, ax, ag)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    print(f"time wgrad n={n} cin={cin} {h}x{w} cout={cout} k={k}: {t*1e6:8.1f} us {2.0*n*h*w*cout*cin*k*k/t/1e12:6.1f} TF", flush=True)

if __name__ == "__main__":
    ok = True
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    if mode in ("all", "check"):
        ok &= check(2, 64, 8, 8, 128, 1)
        ok &= check(3, 96, 14, 14, 128, 1)
        ok &= check(2, 256, 28, 28, 128, 1)
        ok &= check(2, 1024, 7, 7, 512, 1)
        ok &= check(2, 128, 8, 8, 32, 3)
        ok &= check(3, 128, 14, 14, 32, 3)
        ok &= check(2, 128, 56, 56, 32, 3)
        ok &= check(5, 128, 7, 7, 32, 3)
        ok &= check(2, 64, 28, 28, 64, 3)
        ok &= check(3, 40, 9, 11, 24, 3)
        ok &= check(3, 40, 9, 11, 24, 1)
        print("ALL OK" if ok else "SOME FAILED", flush=True)
    if mode in ("all", "time"):
        timeit(42, 256, 56, 56, 128, 1)
        timeit(42, 128, 56, 56, 32, 3)
        timeit(42, 512, 28, 28, 128, 1)
        timeit(42, 128, 28, 28, 32, 3)
        timeit(42, 1024, 14, 14, 128, 1)
        timeit(42, 128, 14, 14, 32, 3)
        timeit(42, 128, 7, 7, 32, 3)
    sys.exit(0 if ok else 1)
