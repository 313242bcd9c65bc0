"""Backward of the head at a given shape: split-fp16 products (bwd_split.hip) against the fp32 matrix-core path and an
fp64 reference; times of forward + backward and of the backward alone.  python tools/bwd_time.py [B N d C] [kind]"""
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nwhead_amd import ops


def ref64(q, s, sy, C, t, kind):
    q = q.double().requires_grad_(True); s = s.double().requires_grad_(True)
    if kind == "euclidean":
        sc = -torch.cdist(q, s)
    elif kind == "cosine":
        sc = F.normalize(q, dim=-1) @ F.normalize(s, dim=-1).t()
    else:
        sc = q @ s.t()
    p = torch.softmax(sc, -1) @ F.one_hot(sy, C).double()
    F.nll_loss(torch.log(p + 1e-12), t).backward()
    return q.grad, s.grad


def run(B, N, d, C, kind, mode):
    os.environ["NW_BWD_SPLIT"] = mode
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    q0, s0 = torch.randn(B, d, generator=g), torch.randn(N, d, generator=g)
    if kind == "dotproduct":
        q0, s0 = q0 * d ** -0.25, s0 * d ** -0.25
    sy = (torch.arange(N) * C // N)
    t = torch.randint(0, C, (B,), generator=g)
    q, s = q0.to(dev).requires_grad_(True), s0.to(dev).requires_grad_(True)
    syd, td = sy.to(dev), t.to(dev)
    out = ops.nw_head(q, s, syd, C, kind)
    loss = F.nll_loss(out, td)
    gq, gs = torch.autograd.grad(loss, (q, s), retain_graph=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(20):
        torch.autograd.grad(loss, (q, s), retain_graph=True)
    e0.record()
    for _ in range(50):
        torch.autograd.grad(loss, (q, s), retain_graph=True)
    e1.record(); torch.cuda.synchronize()
    bwd = e0.elapsed_time(e1) / 50 * 1e3
    e0.record()
    for _ in range(50):
        l = F.nll_loss(ops.nw_head(q, s, syd, C, kind), td)
        torch.autograd.grad(l, (q, s))
    e1.record(); torch.cuda.synchronize()
    both = e0.elapsed_time(e1) / 50 * 1e3
    return gq, gs, bwd, both, (q0, s0, sy, t)


if __name__ == "__main__":
    a = sys.argv[1:]
    B, N, d, C = (int(x) for x in a[:4]) if len(a) >= 4 else (256, 10000, 512, 200)
    kind = a[4] if len(a) > 4 else "euclidean"
    res = {}
    for mode in ("0", "1"):
        gq, gs, bwd, both, data = run(B, N, d, C, kind, mode)
        res[mode] = (gq, gs)
        print(f"NW_BWD_SPLIT={mode}: backward (incl. nll_loss backward) {bwd:.1f} us, forward+backward {both:.1f} us", flush=True)
    q0, s0, sy, t = data
    rq, rs = ref64(q0.cuda(), s0.cuda(), sy.cuda(), C, t.cuda(), kind)
    for mode in ("0", "1"):
        gq, gs = res[mode]
        eq = ((gq.double() - rq).abs().max() / rq.abs().max()).item()
        es = ((gs.double() - rs).abs().max() / rs.abs().max()).item()
        print(f"NW_BWD_SPLIT={mode}: max |err| / max |grad|: gq {eq:.2e}, gs {es:.2e}; finite {bool(torch.isfinite(gq).all() and torch.isfinite(gs).all())}")
