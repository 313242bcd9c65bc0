"""Forward with the bank's cached run tables against the same forward building them per call: bit-equal outputs over
random shapes and label patterns (persistent-kernel shapes), and against the fp32 fallback for sanity."""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd import ops
dev = torch.device("cuda:0")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    B = rng.choice([512, 1000, 2048, 4096])
    N = rng.choice([4099, 10000, 20011, 50000])
    d = rng.choice([96, 128, 512])
    C = rng.choice([1, 2, 37, 200, 1000])
    if B * N < 17_000_000:
        continue
    g = torch.Generator().manual_seed(it)
    q, s = torch.randn(B, d, generator=g).to(dev), torch.randn(N, d, generator=g).to(dev)
    pat = rng.choice(["sorted", "random", "blocks", "const"])
    if pat == "sorted": sy = torch.arange(N) * C // N
    elif pat == "random": sy = torch.randint(0, C, (N,), generator=g)
    elif pat == "blocks": sy = (torch.arange(N) // rng.choice([1, 3, 50, 129])) % C
    else: sy = torch.full((N,), C - 1)
    sy = sy.to(dev)
    plain, tab = ops.SplitBank(s), ops.SplitBank(s, labels=sy)
    a = ops.nw_head(q, s, sy, C, support_cache=plain)
    b = ops.nw_head(q, s, sy, C, support_cache=tab)
    ref = ops.nw_head(q[:64], s, sy, C)          # fp32 path, no cache
    eq = torch.equal(a, b)
    err = (a[:64] - ref).abs().max().item()
    ok = eq and err < 1e-4 and bool(torch.isfinite(a).all())
    bad += not ok
    print(f"{it:3d} B={B:5d} N={N:6d} d={d:4d} C={C:5d} {pat:7s} tables={'yes' if tab.tables is not None else 'no '}: equal {eq}, vs fp32 path {err:.1e}{'' if ok else '   <-- CHECK'}", flush=True)
print("bad", bad)
