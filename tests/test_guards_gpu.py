"""Guards and edge cases of the host layer, on the MI355X: a prepared bank is tied to the tensor it was built
from, labels outside [0, C) are refused where the check is free, rows of subnormal magnitude split to finite
halves, NaN scores sort like torch's, and the library's default dispatch (no NW_SPLIT_ALWAYS) is checked too."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_split_bank_is_tied_to_its_tensor(dev):
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(0)
    s = torch.randn(300, 64, generator=g).to(dev)
    sy = (torch.arange(300) % 7).to(dev)
    q = torch.randn(9, 64, generator=g).to(dev)
    bank = ops.SplitBank(s, sy)
    ops.nw_head(q, s, sy, 7, support_cache=bank)
    with pytest.raises(ValueError, match="another support tensor"):
        ops.nw_head(q, torch.randn_like(s), sy, 7, support_cache=bank)        # same shape, other data
    s.mul_(2.0)                                                               # updated in place
    with pytest.raises(ValueError, match="another support tensor"):
        ops.nw_head(q, s, sy, 7, support_cache=bank)
    with pytest.raises(ValueError, match="outside"):
        ops.nw_head(q, s, sy, 5, support_cache=ops.SplitBank(s, sy))         # label 6 with 5 classes
    with pytest.raises(ValueError, match="non-negative"):
        ops.SplitBank(s, sy - 1)


def test_nwnet_refreshes_stale_inference_state(dev):
    """Weights written while the featurizer stays in eval mode (frozen-BN fine-tuning, load_state_dict) rebuild the
    folded inference copy; a bank swapped in by hand gets a fresh SplitBank; precompute() drops an old shard."""
    import torch.nn as nn
    from nwhead_amd.nwhead.nw import NWNet

    class DS(torch.utils.data.Dataset):
        def __init__(self):
            g = torch.Generator().manual_seed(1)
            self.data, self.targets = torch.randn(60, 3, 8, 8, generator=g), (torch.arange(60) % 4).tolist()

        def __len__(self):
            return 60

        def __getitem__(self, i):
            return self.data[i], self.targets[i]
    feat = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.AdaptiveAvgPool2d(1), nn.Flatten())
    net = NWNet(feat, 4, support_dataset=DS(), n_shot_full=10, device="cuda:0").to(dev).eval()
    net.enable_bn_folding(True)
    net.precompute()
    x = torch.randn(5, 3, 8, 8, generator=torch.Generator().manual_seed(2)).to(dev)
    with torch.no_grad():
        a = net.predict(x, "full")
        with torch.no_grad():
            net.featurizer[0].weight.mul_(1.5)                # in-place update, still in eval mode
        b = net.predict(x, "full")
        want = net.nwhead(net.featurizer(x), net.full_feat, net.full_y)
    assert not torch.allclose(a, b)
    np.testing.assert_allclose(b.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=3e-5)
    net.sharded_bank = object()
    net.precompute()
    assert net.sharded_bank is None
    net.support_eval.full_feat = net.full_feat = net.full_feat * 0.5          # replaced by hand
    with torch.no_grad():
        c = net.predict(x, "full")
        want = net.nwhead(net.featurizer(x), net.full_feat, net.full_y)
    np.testing.assert_allclose(c.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=3e-5)


def test_split_rows_of_subnormal_magnitude(dev):
    from nwhead_amd import ops
    s = torch.zeros(6, 64, device=dev)
    s[0] = 1e-40                      # subnormal
    s[1, 3] = 1e-38
    s[2] = 3e-39
    s[3] = 1.0
    s[4, 0] = float(np.float32(2.0) ** -149)
    bank = ops.SplitBank(s)
    assert torch.isfinite(bank.split.view(torch.float16).float()).all()
    assert torch.isfinite(bank.scale).all() and (bank.scale > 0).all()
    q = torch.randn(4, 64, device=dev)
    sy = torch.arange(6, device=dev) % 2
    s2 = torch.cat([s] * 8)           # N > 25: the fused path
    out = ops.nw_head(q, s2, torch.cat([sy] * 8), 2, support_cache=ops.SplitBank(s2))
    ref = ops.nw_head(q, s2, torch.cat([sy] * 8), 2)
    assert torch.isfinite(out).all()
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=3e-5)


def test_topk_orders_nan_like_torch(dev):
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(3)
    sc = torch.randn(4, 500, generator=g)
    sc[0, 7] = float("nan")
    sc[0, 300] = -float("nan")        # sign bit set
    sc[1, :] = float("nan")
    sc[2, 11] = float("inf")
    sc[2, 12] = float("nan")
    want = torch.argsort(sc, dim=-1, descending=True, stable=True)[:, :16]
    got = ops.nw_topk(sc.to(dev), 16).cpu()
    assert torch.equal(got, want)


def test_default_dispatch_without_split_always():
    """conftest pins NW_SPLIT_ALWAYS=1 for the suite; the library's own choice (fp32 matrix cores with cached norms
    below 2e8 multiply-adds, split-fp16 above) gets a process of its own and the same parity bar."""
    code = r'''
import numpy as np, torch, sys
sys.path.insert(0, %r)
from nwhead_amd import ops
from oracle import nw_oracle as O
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for B, N, d, C in ((64, 1000, 512, 200), (8, 64, 128, 10), (256, 10000, 512, 200), (100, 3000, 96, 17)):
    q, s = torch.randn(B, d, generator=g), torch.randn(N, d, generator=g)
    sy = (torch.arange(N) %% C).sort().values
    sd, syd = s.to(dev), sy.to(dev)
    out = ops.nw_head(q.to(dev), sd, syd, C)
    outc = ops.nw_head(q.to(dev), sd, syd, C, support_cache=ops.SplitBank(sd, syd))
    ref = O.nw_head_f64(q[:32], s, sy, C)
    np.testing.assert_allclose(out[:32].cpu().numpy(), ref.numpy(), rtol=1e-5, atol=3e-5)
    np.testing.assert_allclose(outc[:32].cpu().numpy(), ref.numpy(), rtol=1e-5, atol=3e-5)
print("ok")
''' % ROOT
    env = {k: v for k, v in os.environ.items() if k != "NW_SPLIT_ALWAYS"}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs two devices (one rank per GPU)")
def test_rccl_two_ranks_sharded_predict():
    """The `nccl` (= RCCL) path of ShardedBank.predict_stream on two devices against the oracle: so that the first
    multi-GPU bench run is not also the first RCCL run.  Skipped on the one-GPU test box (the gloo rehearsal on
    one device, test_sharded_hip_world2_gpu.py, covers the same code with another backend)."""
    worker = os.path.join(ROOT, "tests", "_rccl_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", worker],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert r.stdout.count("rccl-ok") == 2


def test_custom_kernel_module_runs_through_the_aggregation_kernels(dev):
    """The reference's NWHead takes any callable kernel (nw.py:256-264).  A module that is not one of the built-in
    score functions computes its scores with torch ops on the device; softmax, label aggregation and log -- and
    their gradient -- run in HIP (nw_aggregate_f32 / nw_aggregate_bwd_f32)."""
    import torch.nn as nn
    import torch.nn.functional as F
    from nwhead_amd.nwhead.nw import NWHead

    class L1(nn.Module):
        def forward(self, x, y):
            return -torch.cdist(x, y, p=1.0)
    g = torch.Generator().manual_seed(0)
    for B, N, d, C, batched in ((7, 40, 16, 5, False), (6, 12, 8, 4, True), (300, 3000, 8, 11, False)):
        x0 = torch.randn(B, d, generator=g)
        s0 = torch.randn(B, N, d, generator=g) if batched else torch.randn(N, d, generator=g)
        sy = torch.randint(0, C, (B, N) if batched else (N,), generator=g)
        t = torch.randint(0, C, (B,), generator=g)
        x64, s64 = x0.double().requires_grad_(True), s0.double().requires_grad_(True)
        sc = -torch.cdist(x64[:, None], s64 if batched else s64[None].expand(B, N, d), p=1.0).squeeze(1)
        oh = F.one_hot(sy, C).double()
        p = torch.einsum("bn,bnc->bc", sc.softmax(-1), oh) if batched else sc.softmax(-1) @ oh
        ref = torch.log(p + 1e-12)
        F.nll_loss(ref, t).backward()
        x, s = x0.to(dev).requires_grad_(True), s0.to(dev).requires_grad_(True)
        out = NWHead(L1(), C)(x, s, sy.to(dev))
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=3e-5)
        F.nll_loss(out, t.to(dev)).backward()
        for got, want in ((x.grad, x64.grad), (s.grad, s64.grad)):
            scale = max(float(want.abs().max()), 1e-3)
            np.testing.assert_allclose(got.cpu().numpy() / scale, want.numpy() / scale, rtol=1e-4, atol=1e-4)


def test_cached_run_tables_give_the_same_output_and_follow_the_labels(dev):
    """A SplitBank built with labels keeps the bank's run tables (nw_bank_tables_build) and names them for the large
    launches (persistent kernel); the output equals the one with tables built inside the call, bit for bit; labels
    modified in place afterwards are no longer the tables' labels: the hint is not given and the result follows them."""
    from nwhead_amd import ops
    torch.manual_seed(0)
    B, N, d, C = 2048, 20000, 128, 50
    q, s = torch.randn(B, d, device=dev), torch.randn(N, d, device=dev)
    sy = (torch.arange(N, device=dev) * C // N)
    plain = ops.SplitBank(s)                      # no labels: tables are built in every call
    withtab = ops.SplitBank(s, labels=sy)
    assert withtab.tables is not None
    a = ops.nw_head(q, s, sy, C, support_cache=plain)
    b = ops.nw_head(q, s, sy, C, support_cache=withtab)
    assert torch.equal(a, b)
    sy[: N // 2] = sy[: N // 2].flip(0)           # in place: another labelling (still valid classes)
    a2 = ops.nw_head(q, s, sy, C, support_cache=plain)
    b2 = ops.nw_head(q, s, sy, C, support_cache=withtab)
    assert torch.equal(a2, b2) and not torch.equal(a, a2)


def test_run_tables_refuse_labels_outside_the_class_range(dev):
    """A bank built WITHOUT labels whose tables are built afterwards (what ShardedBank does) still refuses a call whose
    n_classes does not cover the labels in those tables (they are indexed by class in the merge)."""
    from nwhead_amd import ops
    from nwhead_amd.sharded import ShardedBank
    g = torch.Generator().manual_seed(3)
    s = torch.randn(1024, 64, generator=g).to(dev)
    sy = (torch.arange(1024) % 9).sort().values.to(dev)
    q = torch.randn(16, 64, generator=g).to(dev)
    bank = ops.SplitBank(s)
    bank.build_tables(sy)
    assert bank.tables_label_max == 8
    ops.nw_partials(q, s, sy, 9, support_cache=bank)
    with pytest.raises(ValueError, match="outside"):
        ops.nw_partials(q, s, sy, 5, support_cache=bank)
    with pytest.raises(ValueError, match="non-negative"):
        ops.SplitBank(s).build_tables(sy - 1)
    with pytest.raises(ValueError, match="outside"):
        ShardedBank(s, sy, 5).predict(q)


def test_banks_work_under_inference_mode(dev):
    """torch.inference_mode tensors carry no version counter: the identity signature copes."""
    from nwhead_amd import ops
    from oracle import nw_oracle as O
    g = torch.Generator().manual_seed(4)
    sc, qc = torch.randn(700, 64, generator=g), torch.randn(12, 64, generator=g)
    syc = (torch.arange(700) % 11).sort().values
    with torch.inference_mode():
        s, q, sy = sc.to(dev), qc.to(dev), syc.to(dev)
        bank = ops.SplitBank(s, sy)
        out = ops.nw_head(q, s, sy, 11, support_cache=bank)
        assert bank.matches(s)
    ref = O.nw_head_f64(qc, sc, syc, 11)
    assert (out.cpu().double() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("d", [96, 160])
def test_bank_backward_at_widths_that_are_not_multiples_of_64(dev, d):
    """nw_bwd_bank_f32 with a bank whose split buffer is exactly N * d floats (no tail): the product kernel's last
    64-column tile must not read past it -- the rows are re-split into the workspace at such widths."""
    from nwhead_amd import ops, _lib
    from oracle import nw_oracle as O
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    B, N, C = 64, 1536, 12
    if not _lib.load().nw_bwd_uses_split(B, N, d, C, 0):
        pytest.skip("the split backward does not apply at this shape")
    qc, sc = torch.randn(B, d, generator=g), torch.randn(N, d, generator=g)
    syc, t = (torch.arange(N) % C), torch.randint(0, C, (B,), generator=g)
    q, s = qc.to(dev).requires_grad_(True), sc.to(dev).requires_grad_(True)
    out = ops.nw_head(q, s, syc.to(dev), C)
    F.nll_loss(out, t.to(dev)).backward()
    torch.cuda.synchronize()
    gout = torch.zeros(B, C, dtype=torch.float64)
    gout[torch.arange(B), t] = -1.0 / B
    gq, gs = O.nw_head_bwd_f64(qc, sc, syc, C, gout)
    assert (q.grad.cpu().double() - gq).abs().max().item() < 2e-5 * gq.abs().max().item() + 1e-9
    assert (s.grad.cpu().double() - gs).abs().max().item() < 2e-5 * gs.abs().max().item() + 1e-9


class _TinyDS(torch.utils.data.Dataset):
    def __init__(self, n=60, c=4, hw=8):
        g = torch.Generator().manual_seed(1)
        self.data, self.targets = torch.randn(n, 3, hw, hw, generator=g), (torch.arange(n) % c).tolist()

    def __len__(self):
        return len(self.targets)

    def __getitem__(self, i):
        return self.data[i], self.targets[i]


def test_eval_mode_step_of_the_device_optimizer_refreshes_inference_state(dev):
    """nw_sgd_step_f32 writes parameters through raw pointers; nwhead_amd.optim.SGD must advance their version
    counters, or frozen-BatchNorm fine-tuning (eval mode, grads on, step(), predict()) serves the stale folded copy."""
    import torch.nn as nn
    import torch.nn.functional as F
    from nwhead_amd.nwhead.nw import NWNet
    from nwhead_amd.optim import SGD
    feat = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.AdaptiveAvgPool2d(1), nn.Flatten())
    net = NWNet(feat, 4, support_dataset=_TinyDS(), n_shot_full=10, device="cuda:0").to(dev).eval()
    net.enable_bn_folding(True)
    net.precompute()
    x = torch.randn(5, 3, 8, 8, generator=torch.Generator().manual_seed(2)).to(dev)
    with torch.no_grad():
        a = net.predict(x, "full")
    opt = SGD(net.parameters(), lr=0.5, momentum=0.9, nesterov=True)
    w = net.featurizer[0].weight
    v0 = w._version
    out = net.nwhead(net.featurizer(x), net.full_feat, net.full_y)        # eval mode, gradients on
    F.nll_loss(out, torch.tensor([0, 1, 2, 3, 0], device=dev)).backward()
    opt.step()
    assert w._version > v0 and opt.state[w]["momentum_buffer"]._version > 0
    with torch.no_grad():
        b = net.predict(x, "full")
        want = net.nwhead(net.featurizer(x), net.full_feat, net.full_y)   # the unfolded featurizer with the new weights
    assert not torch.allclose(a, b)
    np.testing.assert_allclose(b.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=3e-5)
    # autograd's saved-tensor check sees the step too: stepping between forward and backward is refused, as with torch's SGD
    out = net.nwhead(net.featurizer(x), net.full_feat, net.full_y)
    opt.step()
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        out.sum().backward()


def test_folded_resnet_predict_under_inference_mode(dev):
    """a folded copy built inside torch.inference_mode() holds inference tensors, which have no version
    counter; ConvBiasAct._split_weight and NWNet._weights_signature must not read `_version` from them."""
    from nwhead_amd.model import load_model
    from nwhead_amd.nwhead.nw import NWNet
    torch.manual_seed(0)
    net = NWNet(load_model("resnet18"), 4, support_dataset=_TinyDS(40, 4, 64), feat_dim=512, n_shot_full=10,
                device="cuda:0").to(dev).eval()
    net.enable_bn_folding(True)
    x = torch.randn(3, 3, 64, 64, generator=torch.Generator().manual_seed(3)).to(dev)
    with torch.inference_mode():
        net.precompute()
        a = net.predict(x, "full")
        b = net.predict(x, "full")                                        # second call: the cached keys are compared
    with torch.no_grad():
        want = net.nwhead(net.featurizer(x), net.full_feat.clone(), net.full_y.clone())
    assert torch.equal(a, b)
    np.testing.assert_allclose(a.cpu().numpy(), want.cpu().numpy(), rtol=2e-4, atol=2e-4)


def test_run_tables_of_another_label_array_are_not_used(dev):
    """nw_fwd_opts.tables are used only for the label array and row count they were built from
    (tables_sy / tables_N); handed the tables of ANOTHER labelling with the same N, the library builds its own."""
    import ctypes as C
    from nwhead_amd import _lib, ops
    torch.manual_seed(1)
    B, N, d, nc = 2048, 20000, 128, 50
    q, s = torch.randn(B, d, device=dev), torch.randn(N, d, device=dev)
    sy = (torch.arange(N, device=dev) * nc // N)
    other = sy.flip(0).contiguous()                       # a different labelling, same N
    bank = ops.SplitBank(s, labels=sy)
    want = ops.nw_head(q, s, sy, nc, support_cache=bank)
    want_other = ops.nw_head(q, s, other, nc, support_cache=ops.SplitBank(s))
    assert not torch.equal(want, want_other)
    lib = _lib.load()
    ws_bytes = lib.nw_fwd_workspace_bytes(B, N, d, nc)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def call(labels, tables_sy, tables_n):
        out = torch.empty(B, nc, device=dev)
        o = _lib.fwd_opts(bank.tables.data_ptr(), bank.tables.numel(), 0, tables_sy, tables_n)
        o.force_split = 1
        rc = lib.nw_fwd_f32(q.data_ptr(), s.data_ptr(), labels.data_ptr(), bank.norm2.data_ptr(), bank.split.data_ptr(),
                            bank.scale.data_ptr(), out.data_ptr(), None, None, None, ws.data_ptr(), ws_bytes, B, N, d, nc, 0,
                            None, 0, 0, C.addressof(o), st)
        assert rc == 0
        torch.cuda.synchronize()
        return out
    assert torch.equal(call(sy, sy.data_ptr(), N), want)                      # the tables' own labels: used
    assert torch.equal(call(other, sy.data_ptr(), N), want_other)             # another array: ignored, own tables built
    assert torch.equal(call(other, other.data_ptr(), N - 1), want_other)      # another row count: ignored
    assert torch.equal(call(other, None, -1), want_other)                     # identity not given: ignored


def test_labels_outside_the_class_range_on_the_unbanked_path(dev):
    """the reference's F.one_hot (nw.py:276) raises for a support label >= n_classes; nw_head without a
    bank skips such supports unless validate_labels is set (NWHead.validate_labels, on under NWNet(debug_mode=True)), in which
    case it raises F.one_hot's RuntimeError."""
    import torch.nn.functional as F
    from nwhead_amd import ops
    from nwhead_amd.nwhead.kernel import get_kernel
    from nwhead_amd.nwhead.nw import NWHead
    g = torch.Generator().manual_seed(9)
    q, s = torch.randn(6, 32, generator=g).to(dev), torch.randn(40, 32, generator=g).to(dev)
    sy = (torch.arange(40) % 5).to(dev)
    bad = sy.clone()
    bad[7] = 5
    with pytest.raises(RuntimeError, match="smaller than num_classes"):
        F.one_hot(bad.cpu(), 5)                                           # what the reference does with it
    with pytest.raises(RuntimeError, match="smaller than num_classes"):
        ops.nw_head(q, s, bad, 5, validate_labels=True)
    with pytest.raises(RuntimeError, match="non-negative"):
        ops.nw_head(q, s, bad - 6, 5, validate_labels=True)
    head = NWHead(get_kernel("euclidean"), 5, validate_labels=True)
    with pytest.raises(RuntimeError, match="smaller than num_classes"):
        head(q, s, bad)
    assert torch.equal(head(q, s, sy), ops.nw_head(q, s, sy, 5))
    # without the check: support 7 is skipped (its weight reaches no class), everything else as if it were absent
    keep = torch.ones(40, dtype=torch.bool, device=dev)
    keep[7] = False
    out = ops.nw_head(q, s, bad, 5)
    assert torch.isfinite(out).all()
    w = torch.softmax(-torch.cdist(q, s), -1)
    want = torch.log(torch.stack([(w * ((bad == c) & keep)).sum(-1) for c in range(5)], -1) + 1e-12)
    np.testing.assert_allclose(out.cpu().numpy(), want.cpu().numpy(), rtol=1e-4, atol=1e-5)
