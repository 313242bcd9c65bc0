"""GPU parity on RANDOM shapes: every draw picks (B, N, d, C), a score kind, a label pattern, shared or
per-query supports and whether the bank is prepared (SplitBank), then holds the C-ABI forward (and, in the
second test, the backward) to the fp64 oracle.  The draws are seeded: a failure names its index.

Same tolerance as test_hip_parity.py (BASELINE.json north_star: 1e-5 relative on the log-probabilities,
absolute floor 3e-5); gradients to 1e-4 of the gradient's own scale.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

KINDS = ("euclidean", "hypersphere_euclidean", "cosine", "dotproduct", "clip")
DIMS = (4, 7, 8, 12, 13, 20, 32, 36, 64, 96, 100, 128, 160, 256, 512)
LS0 = float(np.log(1 / 0.07))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "run with -m gpu on the MI355X box"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from nwhead_amd import ops as o
    return o


@pytest.fixture(scope="module")
def O():
    from oracle import nw_oracle
    return nw_oracle


def _draw(i, max_b, max_n, dims=DIMS):
    r = np.random.RandomState(1000 + i)
    kind = KINDS[r.randint(len(KINDS))]
    d = int(dims[r.randint(len(dims))])
    B = int(r.randint(1, max_b + 1))
    N = int(np.exp(r.uniform(0, np.log(max_n))))              # log-uniform: small banks are the ragged cases
    C = int(r.choice([1, 2, 3, 10, 37, 200, 300]))
    B = max(1, min(B, 60_000_000 // (N * d)))                  # the fp64 oracle materialises (B, N, d) differences
    batched = bool(r.rand() < 0.2) and B * N * d <= 2_000_000   # per-query supports (B,N,d)
    pattern = ("random", "sorted", "single", "sparse")[r.randint(4)]
    g = torch.Generator().manual_seed(5000 + i)
    relu_like = bool(r.rand() < 0.3)                           # post-ReLU-like features: large common mean
    mk = (lambda *sh: torch.rand(*sh, generator=g) * 2) if relu_like else (lambda *sh: torch.randn(*sh, generator=g))
    q = mk(B, d)
    s = mk(B, N, d) if batched else mk(N, d)
    if kind == "dotproduct":                                   # keep |score| O(1..10): the softmax of raw dot
        q, s = q * d ** -0.25, s * d ** -0.25                  # products of 512-dim vectors is a one-hot otherwise
    shape = (B, N) if batched else (N,)
    if pattern == "single":
        sy = torch.full(shape, int(r.randint(C)), dtype=torch.int64)
    elif pattern == "sparse":                                  # most classes absent: log(1e-12) entries
        sy = torch.randint(0, max(1, C // 8), shape, generator=g)
    else:
        sy = torch.randint(0, C, shape, generator=g)
        if pattern == "sorted":
            sy = sy.sort(dim=-1).values
    return dict(kind=kind, B=B, N=N, d=d, C=C, batched=batched, pattern=pattern, q=q, s=s, sy=sy,
                cache=(not batched) and bool(r.rand() < 0.4))


@pytest.mark.parametrize("i", range(48))
def test_forward_random_shapes(dev, ops, O, i):
    c = _draw(i, max_b=300, max_n=6000)
    ls = torch.tensor(LS0, device=dev) if c["kind"] == "clip" else None
    q, s, sy = c["q"].to(dev), c["s"].to(dev), c["sy"].to(dev)
    cache = ops.SplitBank(s, sy) if c["cache"] else None
    out = ops.nw_head(q, s, sy, c["C"], c["kind"], ls, support_cache=cache)
    ref = O.nw_head_f64(c["q"], c["s"], c["sy"], c["C"], c["kind"], LS0)
    assert out.shape == (c["B"], c["C"])
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=3e-5,
                               err_msg=str({k: v for k, v in c.items() if k not in ("q", "s", "sy")}))
    if not c["batched"]:                                       # the sharded form of the same call: partials + merge
        packed = ops.nw_partials(q, s, sy, c["C"], c["kind"], ls, support_cache=cache)
        got = ops.nw_merge(packed.view(1, -1), c["B"], c["C"])
        np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-5, atol=3e-5)


@pytest.mark.parametrize("i", range(8))
def test_forward_random_large_banks(dev, ops, O, i):
    """Prepared banks large enough for the persistent tile kernel (>= 4 tiles per CU), ragged in every
    dimension; a random subset of the query rows is held to the fp64 oracle (rows are independent)."""
    r = np.random.RandomState(77 + i)
    kind = KINDS[i % len(KINDS)]
    d = int(r.choice([96, 128, 224, 256, 512]))
    B, N, C = int(r.randint(1100, 5000)), int(r.randint(9000, 40000)), int(r.choice([2, 10, 200, 1000]))
    g = torch.Generator().manual_seed(900 + i)
    q, s = torch.randn(B, d, generator=g), torch.randn(N, d, generator=g)
    if kind == "dotproduct":
        q, s = q * d ** -0.25, s * d ** -0.25
    sy = torch.randint(0, C, (N,), generator=g)
    if i % 2 == 0:
        sy = sy.sort().values
    ls = torch.tensor(LS0, device=dev) if kind == "clip" else None
    qd, sd, syd = q.to(dev), s.to(dev), sy.to(dev)
    out = ops.nw_head(qd, sd, syd, C, kind, ls, support_cache=ops.SplitBank(sd, syd))
    rows = torch.from_numpy(r.choice(B, 8, replace=False))
    ref = O.nw_head_f64(q[rows], s, sy, C, kind, LS0)
    np.testing.assert_allclose(out[rows.to(dev)].cpu().numpy(), ref.numpy(), rtol=1e-5, atol=3e-5,
                               err_msg=f"{kind} B={B} N={N} d={d} C={C}")
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("i", range(20))
def test_backward_random_shapes(dev, ops, O, i):
    c = _draw(100 + i, max_b=48, max_n=700, dims=(4, 8, 13, 20, 32, 64, 100, 128))
    B, C, kind = c["B"], c["C"], c["kind"]
    t = torch.randint(0, C, (B,), generator=torch.Generator().manual_seed(i))
    # fp64 autograd through the oracle head
    q64 = c["q"].double().requires_grad_(True)
    s64 = c["s"].double().requires_grad_(True)
    ls64 = torch.tensor(LS0, dtype=torch.float64, requires_grad=True)
    F.nll_loss(O.nw_head_f64(q64, s64, c["sy"], C, kind, ls64) if kind != "clip" else
               _clip_head_f64(q64, s64, c["sy"], C, ls64), t).backward()
    q = c["q"].to(dev).requires_grad_(True)
    s = c["s"].to(dev).requires_grad_(True)
    ls = torch.tensor(LS0, device=dev, requires_grad=True) if kind == "clip" else None
    F.nll_loss(ops.nw_head(q, s, c["sy"].to(dev), C, kind, ls), t.to(dev)).backward()
    info = str({k: v for k, v in c.items() if k not in ("q", "s", "sy")})
    for got, ref in ((q.grad, q64.grad), (s.grad, s64.grad)):
        ref = ref.numpy()
        # an all-one-class support has exactly zero gradient: fp32 leaves the rounding of its O(1/B) terms
        scale = max(float(np.abs(ref).max()), 1e-3)
        np.testing.assert_allclose(got.cpu().numpy() / scale, ref / scale, rtol=1e-4, atol=1e-4, err_msg=info)
    if kind == "clip":
        np.testing.assert_allclose(ls.grad.item(), ls64.grad.item(), rtol=1e-4, atol=1e-6, err_msg=info)


@pytest.mark.parametrize("B,N,d,C", [(64, 1000, 512, 200), (200, 1037, 96, 10), (1000, 150, 64, 5), (37, 4099, 32, 3),
                                       (129, 257, 132, 7)])
@pytest.mark.parametrize("kind", KINDS)
def test_backward_matrix_core_path(dev, ops, O, monkeypatch, B, N, d, C, kind):
    """Shapes past the B*N*d >= 2^22 threshold: both products of the backward run in nw_bwd_gemm_kernel
    (ragged M, N and K; K split over workgroups for the tall and the wide case).  NW_BWD_SPLIT=0 keeps the shapes
    with d % 32 == 0 on this fp32 path (by default they take the split-row products, tested below)."""
    monkeypatch.setenv("NW_BWD_SPLIT", "0")
    g = torch.Generator().manual_seed(B + N)
    q0, s0 = torch.randn(B, d, generator=g), torch.randn(N, d, generator=g)
    if kind == "dotproduct":
        q0, s0 = q0 * d ** -0.25, s0 * d ** -0.25
    sy = torch.randint(0, C, (N,), generator=g)
    t = torch.randint(0, C, (B,), generator=g)
    q64, s64 = q0.double().requires_grad_(True), s0.double().requires_grad_(True)
    ls64 = torch.tensor(LS0, dtype=torch.float64, requires_grad=True)
    F.nll_loss(O.nw_head_f64(q64, s64, sy, C, kind, ls64) if kind != "clip" else _clip_head_f64(q64, s64, sy, C, ls64),
               t).backward()
    q, s = q0.to(dev).requires_grad_(True), s0.to(dev).requires_grad_(True)
    ls = torch.tensor(LS0, device=dev, requires_grad=True) if kind == "clip" else None
    F.nll_loss(ops.nw_head(q, s, sy.to(dev), C, kind, ls), t.to(dev)).backward()
    for got, ref in ((q.grad, q64.grad), (s.grad, s64.grad)):
        ref = ref.numpy()
        scale = max(float(np.abs(ref).max()), 1e-3)
        np.testing.assert_allclose(got.cpu().numpy() / scale, ref / scale, rtol=1e-4, atol=1e-4)
    if kind == "clip":
        np.testing.assert_allclose(ls.grad.item(), ls64.grad.item(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("B,N,d,C", [(256, 10000, 512, 200), (200, 1037, 96, 10), (70, 1100, 32, 3), (129, 2500, 160, 7),
                                       (64, 1024, 64, 5), (33, 4099, 64, 1000), (129, 257, 160, 1000)])
@pytest.mark.parametrize("kind", KINDS)
def test_backward_split_fp16_path(dev, ops, O, monkeypatch, B, N, d, C, kind):
    """Both products of the backward on the fp16 matrix cores (bwd_split.hip: split-row operands, transposed LDS reads,
    K split over workgroups for the first product): ragged M, N and K tiles, every kernel type; the same bar as the
    fp32 matrix-core path.  The path is the default from B >= 16, N >= 256, B N d >= 2^22 with d % 32 == 0;
    NW_BWD_SPLIT=1 takes it wherever the shape allows.  The last shape has more classes than supports: most queries'
    target class is carried by no support, their rows of coefficients are exactly zero -- such a row once set the batch's
    scale for the queries' split image (2^0 against 2^-30 for the others) and every other row underflowed."""
    if B < 64:
        monkeypatch.setenv("NW_BWD_SPLIT", "1")   # (the others take the path by default)
    g = torch.Generator().manual_seed(B + N)
    q0, s0 = torch.randn(B, d, generator=g), torch.randn(N, d, generator=g)
    if kind == "dotproduct":
        q0, s0 = q0 * d ** -0.25, s0 * d ** -0.25
    if (B, N) == (129, 2500):
        s0 = s0 * torch.logspace(-3, 3, N).unsqueeze(1) if kind in ("cosine", "hypersphere", "clip") else s0 * 3.0
    sy = torch.randint(0, C, (N,), generator=g)
    t = torch.randint(0, C, (B,), generator=g)
    q64, s64 = q0.double().requires_grad_(True), s0.double().requires_grad_(True)
    ls64 = torch.tensor(LS0, dtype=torch.float64, requires_grad=True)
    F.nll_loss(O.nw_head_f64(q64, s64, sy, C, kind, ls64) if kind != "clip" else _clip_head_f64(q64, s64, sy, C, ls64),
               t).backward()
    q, s = q0.to(dev).requires_grad_(True), s0.to(dev).requires_grad_(True)
    ls = torch.tensor(LS0, device=dev, requires_grad=True) if kind == "clip" else None
    F.nll_loss(ops.nw_head(q, s, sy.to(dev), C, kind, ls), t.to(dev)).backward()
    for got, ref in ((q.grad, q64.grad), (s.grad, s64.grad)):
        ref = ref.numpy()
        assert torch.isfinite(got).all()
        scale = max(float(np.abs(ref).max()), 1e-3)
        np.testing.assert_allclose(got.cpu().numpy() / scale, ref / scale, rtol=1e-4, atol=1e-4)
    if kind == "clip":
        np.testing.assert_allclose(ls.grad.item(), ls64.grad.item(), rtol=1e-4, atol=1e-6)


def test_backward_split_fp16_matches_fp32_matrix_core_path(dev, ops, monkeypatch):
    """The two implementations of the products against each other, tighter than either against fp64 of the whole
    head: 2e-5 of the largest gradient entry (the forwards differ too: fp32 matrix cores against split rows)."""
    B, N, d, C = 256, 10000, 512, 200
    g = torch.Generator().manual_seed(3)
    q0, s0 = torch.randn(B, d, generator=g), torch.randn(N, d, generator=g)
    sy = (torch.arange(N) * C // N).to(dev)
    t = torch.randint(0, C, (B,), generator=g).to(dev)
    grads = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("NW_BWD_SPLIT", mode)
        q, s = q0.to(dev).requires_grad_(True), s0.to(dev).requires_grad_(True)
        F.nll_loss(ops.nw_head(q, s, sy, C, "euclidean"), t).backward()
        grads[mode] = (q.grad.clone(), s.grad.clone())
    for a, b in zip(grads["0"], grads["1"]):
        assert ((a - b).abs().max() / a.abs().max()).item() < 2e-5


def _clip_head_f64(q, s, sy, C, ls):
    """The oracle's fp64 head takes logit_scale by value; this keeps it in the graph (kernel.py:35-44)."""
    qn = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    sn = s / s.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    sc = ls.exp() * (torch.einsum("bd,bnd->bn", qn, sn) if s.dim() == 3 else qn @ sn.t())
    w = torch.softmax(sc, dim=-1)
    oh = F.one_hot(sy, C).double()
    p = torch.einsum("bn,bnc->bc", w, oh) if oh.dim() == 3 else w @ oh
    return torch.log(p + 1e-12)
