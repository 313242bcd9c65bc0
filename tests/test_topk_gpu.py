"""nw_topk_f32 (topk.hip): the first k columns of the reference's full descending argsort
(KNN.__call__, nwhead/utils.py:185-193), bit-exact, ties in ascending index order."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from nwhead_amd import ops
    return ops


def _ref(scores, k):
    return torch.argsort(scores.cpu(), dim=-1, descending=True, stable=True)[:, :k]


@pytest.mark.parametrize("B,N,k", [(1, 1, 1), (3, 5, 5), (7, 64, 10), (16, 1000, 20), (5, 1000, 257),
                                   (4, 4097, 1024), (32, 50000, 10), (2, 50000, 1000), (9, 333, 1)])
def test_topk_matches_stable_argsort(dev, ops, B, N, k):
    g = torch.Generator().manual_seed(B * 131 + N)
    s = (torch.randn(B, N, generator=g) * 3 - 30).to(dev)          # like -distance: one sign, one exponent
    idx, vals = ops.nw_topk(s, k, return_values=True)
    ref = _ref(s, k)
    assert torch.equal(idx.cpu(), ref)
    assert torch.equal(vals.cpu(), torch.gather(s.cpu(), 1, ref))


def test_topk_ties_and_specials(dev, ops):
    g = torch.Generator().manual_seed(2)
    # heavy ties: scores drawn from 7 distinct values; the threshold value is shared by many columns
    s = torch.randint(-3, 4, (6, 2000), generator=g).float()
    s[0, :] = 1.5                                                   # a constant row: the first k indices
    s[1, ::3] = float("inf")
    s[2, 5] = float("nan")                                          # NaN sorts first, like torch
    s[3, :] = -s[3, :].abs()
    s[4, :] = -s[4, :].abs() - 1.0
    s[4, 40:90:2] = -0.0                                            # -0.0 and +0.0 are equal for torch:
    s[4, 41:91:2] = 0.0                                             # the zeros come out in index order
    s = s.to(dev)
    for k in (1, 8, 100, 777):
        assert torch.equal(ops.nw_topk(s, k).cpu(), _ref(s, k)), k


def test_topk_mixed_signs_and_wide_range(dev, ops):
    g = torch.Generator().manual_seed(3)
    s = (torch.randn(8, 30000, generator=g) * torch.logspace(-20, 20, 30000)).to(dev)
    assert torch.equal(ops.nw_topk(s, 64).cpu(), _ref(s, 64))


def test_topk_rejects_large_k(dev, ops):
    from nwhead_amd._lib import NWHipError
    with pytest.raises(NWHipError):
        ops.nw_topk(torch.randn(2, 5000, device=dev), 1025)
    with pytest.raises(NWHipError):
        ops.nw_topk(torch.randn(2, 8, device=dev), 9)


def test_knn_support_uses_topk(dev, ops):
    """KNN (the 'knn' / 'hnsw' support modes): same rows as the reference's argsort-and-slice."""
    from nwhead_amd.nwhead.utils import KNN
    g = torch.Generator().manual_seed(4)
    bank = torch.randn(3000, 64, generator=g).to(dev)
    bank[100] = bank[7]                                            # duplicate rows: exact score ties
    labels = (torch.arange(3000) % 10).to(dev)
    q = torch.cat([bank[7:8] + 0.0, torch.randn(5, 64, generator=g).to(dev)])
    knn = KNN(bank, labels, n_neighbors=20)
    idx = knn.indices(q)
    scores = ops.nw_scores(q, bank, "euclidean")
    assert torch.equal(idx.cpu(), _ref(scores, 20))
    sx, sy = knn(q)
    assert sx.shape == (6 * 20, 64) and torch.equal(sy.cpu(), labels.cpu()[idx.cpu().reshape(-1)])


def test_scores_through_the_bank_and_few_class_merge():
    """ops.nw_scores(..., support_cache=bank): the neighbour search's (B, N) score matrix from the split-fp16 tile kernel
    (the forward with its score output and one class) against the fp32 scores kernel and fp64; the merge behind it sums
    ONE class over every tile (the lanes of a group share the tiles), so C = 1 and C = 2 outputs are checked too."""
    from nwhead_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    for B, N, d in ((256, 10000, 512), (37, 4100, 96)):
        q, s = torch.randn(B, d, generator=g).to(dev), torch.randn(N, d, generator=g).to(dev)
        bank = ops.SplitBank(s)
        a = ops.nw_scores(q, s)
        b = ops.nw_scores(q, s, support_cache=bank)
        ref = -torch.cdist(q.double(), s.double())
        assert (a.double() - ref).abs().max().item() < 3e-5 and (b.double() - ref).abs().max().item() < 3e-5
        for C in (1, 2):
            sy = (torch.arange(N, device=dev) * C // N)
            out = ops.nw_head(q, s, sy, C, support_cache=bank)
            w = torch.softmax(ref, -1)
            want = torch.log(torch.stack([w[:, sy == c].sum(1) for c in range(C)], 1) + 1e-12)
            assert (out.double() - want).abs().max().item() < 3e-5


def test_many_classes_merge_tables_in_global_memory():
    """More classes than the run merge holds tables for in LDS (C > ~4200): the tables are built once per launch in the
    workspace (nw_class_tables_kernel).  Outputs against fp64, sorted and shuffled labels, with and without a bank."""
    from nwhead_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(12)
    B, N, d, C = 192, 12000, 64, 6000
    q, s = torch.randn(B, d, generator=g).to(dev), torch.randn(N, d, generator=g).to(dev)
    w = torch.softmax(-torch.cdist(q.double(), s.double()), -1)
    for sy in ((torch.arange(N) * C // N), torch.randint(0, C, (N,), generator=g)):
        syd = sy.to(dev)
        want = torch.log(w @ torch.nn.functional.one_hot(syd, C).double() + 1e-12)
        for cache in (None, ops.SplitBank(s, labels=syd)):
            out = ops.nw_head(q, s, syd, C, support_cache=cache)
            assert (out.double() - want).abs().max().item() < 3e-5


def test_embedding_sizes_that_are_not_multiples_of_four():
    """d % 4 != 0: nw_head pads the operands with zero columns (same scores, same norms) so that the tile kernels serve
    them; outputs and gradients against fp64, with and without a bank's cached norms."""
    import torch.nn.functional as F
    from nwhead_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(13)
    for kind in ("euclidean", "cosine"):
        for B, N, d, C in ((33, 700, 130, 10), (64, 3000, 67, 50)):
            q0, s0 = torch.randn(B, d, generator=g), torch.randn(N, d, generator=g)
            sy = torch.randint(0, C, (N,), generator=g)
            t = torch.randint(0, C, (B,), generator=g)
            q64, s64 = q0.double().requires_grad_(True), s0.double().requires_grad_(True)
            if kind == "euclidean":
                sc = -torch.cdist(q64, s64)
            else:
                sc = F.normalize(q64, dim=-1) @ F.normalize(s64, dim=-1).t()
            ref = torch.log(torch.softmax(sc, -1) @ F.one_hot(sy, C).double() + 1e-12)
            F.nll_loss(ref, t).backward()
            q, s = q0.to(dev).requires_grad_(True), s0.to(dev).requires_grad_(True)
            out = ops.nw_head(q, s, sy.to(dev), C, kind)
            F.nll_loss(out, t.to(dev)).backward()
            assert (out.detach().cpu().double() - ref.detach()).abs().max().item() < 3e-5
            for got, want in ((q.grad, q64.grad), (s.grad, s64.grad)):
                assert got.shape == want.shape
                assert ((got.cpu().double() - want).abs().max() / want.abs().max()).item() < 1e-4
            sd = s0.to(dev)
            bank = ops.SplitBank(sd)
            out2 = ops.nw_head(q0.to(dev), sd, sy.to(dev), C, kind, support_cache=bank)
            assert (out2.cpu().double() - ref.detach()).abs().max().item() < 3e-5


def test_banks_whose_width_is_not_a_multiple_of_32():
    """SplitBank of a (N, d) bank with d % 32 != 0, d >= 64: the bank keeps rows padded with zero columns (split format),
    callers pad the queries: nw_head (sorted and shuffled labels, with weights, with a gradient for the queries),
    nw_scores, nw_head_influence and the sharded partial path against fp64."""
    import torch.nn.functional as F
    from nwhead_amd import ops
    from nwhead_amd.sharded import ShardedBank
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(14)
    for B, N, d, C in ((96, 6000, 100, 20), (300, 20000, 130, 7)):
        q0, s0 = torch.randn(B, d, generator=g), torch.randn(N, d, generator=g)
        q, s = q0.to(dev), s0.to(dev)
        sc64 = -torch.cdist(q0.double(), s0.double())
        w64 = torch.softmax(sc64, -1)
        for sy in ((torch.arange(N) * C // N), torch.randint(0, C, (N,), generator=g)):
            syd = sy.to(dev)
            want = torch.log(w64 @ F.one_hot(sy, C).double() + 1e-12)
            bank = ops.SplitBank(s, labels=syd)
            assert bank.pad == (-d) % 32 and bank.split is not None
            out = ops.nw_head(q, s, syd, C, support_cache=bank)
            assert out.shape == (B, C) and (out.cpu().double() - want).abs().max().item() < 3e-5
            out_w, wts = ops.nw_head(q, s, syd, C, return_weights=True, support_cache=bank)
            assert (wts.cpu().double() - w64).abs().max().item() < 3e-6 and (out_w.cpu().double() - want).abs().max().item() < 3e-5
            qg = q.clone().requires_grad_(True)
            t = torch.randint(0, C, (B,), generator=g)
            F.nll_loss(ops.nw_head(qg, s, syd, C, support_cache=bank), t.to(dev)).backward()
            q64 = q0.double().requires_grad_(True)
            F.nll_loss(torch.log(torch.softmax(-torch.cdist(q64, s0.double()), -1) @ F.one_hot(sy, C).double() + 1e-12), t).backward()
            assert qg.grad.shape == (B, d)
            assert ((qg.grad.cpu().double() - q64.grad).abs().max() / q64.grad.abs().max()).item() < 1e-4
        plain = ops.SplitBank(s)
        assert (ops.nw_scores(q, s, support_cache=plain).cpu().double() - sc64).abs().max().item() < 3e-5
        sy = (torch.arange(N) * C // N); syd = sy.to(dev)
        qy = torch.randint(0, C, (B,), generator=g).to(dev)
        out_i, infl = ops.nw_head_influence(q, s, syd, C, qy, support_cache=plain)
        want = torch.log(w64 @ F.one_hot(sy, C).double() + 1e-12)
        assert (out_i.cpu().double() - want).abs().max().item() < 3e-5 and infl.shape == (B, N)
        sb = ShardedBank(s, syd, C)
        assert sb.cache.pad == (-d) % 32
        got = sb.predict_stream([q[:50], q[50:]], bucket=2)
        assert (torch.cat(got).cpu().double() - want).abs().max().item() < 3e-5
