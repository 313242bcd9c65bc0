"""Random shapes through nw_conv3x3_f32 (every tile form) against torch's conv2d, and nw_topk_f32 against a stable
descending argsort (ties, NaN, -0.0): a wider net than the test suite's fixed cases."""
import os, sys, random
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd import ops, _lib
dev = torch.device("cuda:0")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
lib = _lib.load()
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    n = rng.choice([1, 2, 5, 16, 64, 100])
    cin = rng.choice([3, 8, 20, 64, 128, 250, 520])
    cout = rng.choice([32, 64, 96, 128, 256])
    h, w = rng.choice([(7, 7), (14, 14), (28, 28), (56, 56), (5, 9), (12, 16), (8, 8), (33, 20)])
    if n * cin * cout * h * w * 9 > 6e10:
        continue
    g = torch.Generator().manual_seed(it)
    x = torch.randn(n, cin, h, w, generator=g).to(dev)
    wgt = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).to(dev)
    b = torch.randn(cout, generator=g).to(dev) if rng.random() < 0.5 else None
    r = torch.randn(n, cout, h, w, generator=g).to(dev) if rng.random() < 0.3 else None
    relu = rng.random() < 0.5
    ref = F.conv2d(x.double(), wgt.double(), None if b is None else b.double(), padding=1)
    if r is not None: ref = ref + r.double()
    if relu: ref = F.relu(ref)
    out = ops.conv3x3(x, ops.conv3x3_weight(wgt), cin, b, r, relu)
    err = ((out.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-9)).item()
    wg = lib.nw_conv3x3_workgroups(n, cin, cout, h, w)
    flag = "" if err < 2e-5 else "   <-- CHECK"
    bad += err >= 2e-5
    print(f"conv {it:3d} n={n:3d} {cin:4d}->{cout:<4d} {h:2d}x{w:<2d} wg={wg:5d} ws={lib.nw_conv3x3_workspace_bytes(n, cin, cout, h, w):9d}: {err:.1e}{flag}", flush=True)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    B = rng.choice([1, 3, 64, 257])
    N = rng.choice([1, 2, 63, 64, 65, 1000, 1024, 1025, 4099, 16384, 16385, 20000])
    k = min(N, rng.choice([1, 2, 10, 20, 63, 64, 65, 200, 1000]))
    g = torch.Generator().manual_seed(1000 + it)
    sc = torch.randn(B, N, generator=g)
    mode = rng.choice(["plain", "ties", "nan", "const"])
    if mode == "ties": sc = (sc * 3).round() / 3
    if mode == "nan" and N > 4:
        sc[:, ::7] = float("nan"); sc[:, 1::11] = -0.0; sc[:, 2::13] = 0.0
    if mode == "const": sc = torch.zeros(B, N)
    scd = sc.to(dev)
    idx, val = ops.nw_topk(scd, k, return_values=True)
    want = torch.argsort(scd, dim=-1, descending=True, stable=True)[:, :k]
    ok = torch.equal(idx, want)
    wv = torch.gather(scd, 1, want)
    okv = torch.equal(torch.nan_to_num(val, nan=123.0), torch.nan_to_num(wv, nan=123.0))
    bad += not (ok and okv)
    print(f"topk {it:3d} B={B:3d} N={N:5d} k={k:4d} {mode:5s}: idx {ok} val {okv}{'' if ok and okv else '   <-- CHECK'}", flush=True)
print("bad", bad)
