#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Run only in the build container (needs /root/reference, which never travels
to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py

The reference (alanqrwang/nwhead @ 2024_08_07) is imported unmodified from
/root/reference.  ``nwhead/utils.py:4`` imports ``hnswlib`` which is not
installed; an exact-kNN stand-in module is put in ``sys.modules`` first
 -- it only matters for ``precompute()`` which builds the index
unconditionally (``nwhead/support.py:133``).  Fixtures store inputs AND
outputs so nothing depends on RNG reproducibility across torch builds.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = os.environ.get("NWHEAD_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _install_hnswlib_stub():
    mod = types.ModuleType("hnswlib")

    class Index:  # exact kNN stand-in with the three calls the reference makes
        def __init__(self, space, dim):
            self.dim = dim

        def init_index(self, max_elements, ef_construction=100, M=16):
            pass

        def add_items(self, data):
            self.data = np.asarray(data, dtype=np.float32)

        def knn_query(self, x, k=1):
            x = np.asarray(x, dtype=np.float32)
            d = ((x[:, None, :] - self.data[None]) ** 2).sum(-1)
            idx = np.argsort(d, axis=1, kind="stable")[:, :k]
            return idx, np.take_along_axis(d, idx, 1)

    mod.Index = Index
    sys.modules["hnswlib"] = mod


_install_hnswlib_stub()
sys.path.insert(0, REF)
from nwhead.kernel import get_kernel          # noqa: E402
from nwhead.nw import NWHead, NWNet           # noqa: E402
from util.metric import support_influence     # noqa: E402

KINDS = ["euclidean", "hypersphere_euclidean", "cosine", "dotproduct", "clip"]


def npz(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KB")


def head(kind, C):
    return NWHead(get_kernel(kind), C)


def g1():
    """K1 shapes, all five kernels, 2-D and 3-D support."""
    g = torch.Generator().manual_seed(101)
    B, N, d, C = 8, 64, 128, 10
    x = torch.randn(B, d, generator=g)
    sx = torch.randn(N, d, generator=g)
    sy = torch.randint(0, C, (N,), generator=g)
    sx3 = torch.randn(B, N, d, generator=g)
    sy3 = torch.randint(0, C, (B, N), generator=g)
    out = {"x": x, "sx": sx, "sy": sy, "sx3": sx3, "sy3": sy3, "C": C}
    with torch.no_grad():
        for k in KINDS:
            out[f"out2d_{k}"] = head(k, C)(x, sx, sy)
            out[f"out3d_{k}"] = head(k, C)(x, sx3, sy3)
            if "euclidean" in k:      # 2-D use (nw.py:248) only works for the cdist kernels;
                out[f"scores2d_{k}"] = get_kernel(k)(x, sx)   # torch.bmm raises on 2-D input
    npz("g1_k1_all_kernels.npz", **out)


def g2():
    """cdist regime switch: N=20 (direct), 25, 26 (first mm size)."""
    g = torch.Generator().manual_seed(202)
    d, C, B = 32, 5, 6
    out = {"C": C}
    for N in (20, 25, 26):
        x = torch.randn(B, d, generator=g) * 3 + 1.0
        sx = torch.randn(N, d, generator=g) * 3 + 1.0
        sy = torch.randint(0, C, (N,), generator=g)
        with torch.no_grad():
            o = head("euclidean", C)(x, sx, sy)
        out.update({f"x_{N}": x, f"sx_{N}": sx, f"sy_{N}": sy, f"out_{N}": o})
    npz("g2_cdist_regimes.npz", **out)


def g3():
    """backward of nll_loss(NWHead(...)), 2-D and 3-D support, plus D=0 rows."""
    g = torch.Generator().manual_seed(303)
    B, d, C = 8, 48, 6
    out = {"C": C}
    for tag, N in (("n20", 20), ("n64", 64)):
        x = torch.randn(B, d, generator=g)
        sx = torch.randn(N, d, generator=g)
        sy = torch.randint(0, C, (N,), generator=g)
        t = torch.randint(0, C, (B,), generator=g)
        # D = 0: queries 0 and 3 are exact copies of support rows
        sx[1] = x[0]
        sx[N - 1] = x[3]
        for kind in KINDS:
            xr = x.clone().requires_grad_(True)
            sr = sx.clone().requires_grad_(True)
            h = head(kind, C)
            o = h(xr, sr, sy)
            F.nll_loss(o, t).backward()
            out[f"{tag}_{kind}_out"] = o.detach()
            out[f"{tag}_{kind}_gx"] = xr.grad
            out[f"{tag}_{kind}_gs"] = sr.grad
            if kind == "clip":
                out[f"{tag}_{kind}_gls"] = h.kernel.logit_scale.grad
        out.update({f"{tag}_x": x, f"{tag}_sx": sx, f"{tag}_sy": sy, f"{tag}_t": t})
        # 3-D support
        sx3 = torch.randn(B, N, d, generator=g)
        sy3 = torch.randint(0, C, (B, N), generator=g)
        xr = x.clone().requires_grad_(True)
        sr = sx3.clone().requires_grad_(True)
        o = head("euclidean", C)(xr, sr, sy3)
        F.nll_loss(o, t).backward()
        out.update({f"{tag}_sx3": sx3, f"{tag}_sy3": sy3, f"{tag}_out3": o.detach(),
                    f"{tag}_gx3": xr.grad, f"{tag}_gs3": sr.grad})
    npz("g3_backward.npz", **out)


def g4():
    """support_influence, B=1 and B=4, incl. the 1-shot +inf case and the 3-D quirk shape."""
    g = torch.Generator().manual_seed(404)
    N, d, C = 64, 16, 10
    out = {"C": C}
    for B in (1, 4):
        x = torch.randn(B, d, generator=g)
        sx = torch.randn(N, d, generator=g)
        sy = torch.arange(N) % C
        sy[5] = C - 1                       # perturb the balance
        # make class 0 a one-shot class for query 0 -> denominator 0 -> +inf
        sy[sy == 0] = 1
        sy[0] = 0
        qy = torch.randint(0, C, (B,), generator=g)
        qy[0] = 0
        with torch.no_grad():
            scores = get_kernel("euclidean")(x[:, None], sx[None].expand(B, N, d)).squeeze(1)
            w = F.softmax(scores, -1)
            sm = torch.exp(head("euclidean", C)(x, sx, sy))
            infl = support_influence(sm, F.one_hot(qy, C).float(), w, F.one_hot(sy, C).float())
            quirk = support_influence(sm, F.one_hot(qy, C).float(), w,
                                      F.one_hot(sy, C).float()[None].expand(B, N, C))
        out.update({f"b{B}_softmaxes": sm, f"b{B}_qy": qy, f"b{B}_w": w, f"b{B}_sy": sy,
                    f"b{B}_infl": infl, f"b{B}_quirk_shape": np.array(quirk.shape)})
    npz("g4_support_influence.npz", **out)


class _FakeDS(torch.utils.data.Dataset):
    def __init__(self, n, C, shape, seed):
        g = torch.Generator().manual_seed(seed)
        self.data = torch.randn(n, *shape, generator=g)
        self.targets = torch.randint(0, C, (n,), generator=g).tolist()
        self.num_classes = C

    def __len__(self):
        return len(self.targets)

    def __getitem__(self, i):
        return self.data[i], self.targets[i]


def g5():
    """NWNet plumbing with a tiny featurizer: bank ordering, modes, sampler draws."""
    C, shape, n = 10, (3, 4, 4), 200
    ds = _FakeDS(n, C, shape, seed=505)
    torch.manual_seed(0)
    feat = nn.Sequential(nn.Flatten(), nn.Linear(48, 16))
    net = NWNet(feat, C, support_dataset=ds, feat_dim=16, n_shot=2, n_way=6,
                n_shot_full=7, n_shot_cluster=2, n_neighbors=3, device="cpu")
    net.eval()
    np.random.seed(1234)
    net.precompute()
    g = torch.Generator().manual_seed(506)
    xq = torch.randn(5, *shape, generator=g)
    yq = torch.tensor([2, 6, 0, 3, 3])
    out = {"C": C, "ds_data": ds.data, "ds_targets": np.array(ds.targets),
           "w": feat[1].weight.detach(), "b": feat[1].bias.detach(),
           "full_feat": net.full_feat, "full_y": net.full_y,
           "cluster_feat": net.support_eval.cluster_feat, "cluster_y": net.support_eval.cluster_y,
           "xq": xq, "yq": yq}
    with torch.no_grad():
        for mode in ("full", "cluster", "knn", "hnsw", "ensemble"):
            out[f"pred_{mode}"] = net.predict(xq, mode)
        np.random.seed(77)
        out["pred_random"] = net.predict(xq, "random")
        np.random.seed(77)
        _, ry, _ = net.support_eval.random_iter.next()
        out["random_sy"] = ry
        out["neighbors"] = net.get_neighbors(xq)
        # train-mode forward through the sampler: record what it drew
        np.random.seed(99)
        sx, sy, sm = net.support_train.get_support(yq)
        out["train_sx"] = sx
        out["train_sy"] = sy
        out["fwd_support_data"] = net(xq, yq, support_data=(sx, sy, None))
        np.random.seed(99)
        out["fwd_sampled"] = net(xq, yq)
    npz("g5_nwnet_plumbing.npz", **out)


def g7():
    """shard-merge: K1-scale problem split into 8 shards (reference gives the unsharded answer)."""
    g = torch.Generator().manual_seed(707)
    B, N, d, C = 8, 64, 128, 10
    x = torch.randn(B, d, generator=g)
    sx = torch.randn(N, d, generator=g)
    sy = torch.arange(N) % C
    sy = sy.sort().values                  # class-sorted, balanced like the 'full' bank
    with torch.no_grad():
        o = head("euclidean", C)(x, sx, sy)
    npz("g7_shard_merge.npz", x=x, sx=sx, sy=sy, out=o, C=C, n_shards=8)


def g8():
    """adversarial: large-norm features with small distances; far supports (exp underflow)."""
    g = torch.Generator().manual_seed(808)
    B, N, d, C = 8, 96, 64, 6
    base = torch.randn(1, d, generator=g)
    base = base / base.norm() * 30.0
    x = base + 0.12 * torch.randn(B, d, generator=g)
    sx = base + 0.12 * torch.randn(N, d, generator=g)
    sy = torch.randint(0, C, (N,), generator=g)
    xf = torch.randn(B, d, generator=g)
    sxf = torch.randn(N, d, generator=g) + 14.0    # every distance > 90
    with torch.no_grad():
        o_near = head("euclidean", C)(x, sx, sy)
        o_far = head("euclidean", C)(xf, sxf, sy)
    npz("g8_adversarial.npz", x=x, sx=sx, sy=sy, xf=xf, sxf=sxf, out_near=o_near, out_far=o_far, C=C)
    # (the reference cannot run in float64: nw.py:276 casts the one-hot to float32 and bmm then
    #  rejects mixed dtypes -- the fp64 "true values" come from oracle/nw_oracle.py instead)


def g6():
    """Backbones with procedural weights (tests/procedural.py): eval and train-mode outputs."""
    sys.path.insert(0, os.path.dirname(OUT))
    from procedural import fill_procedural
    from model.resnet import resnet18, CIFAR_ResNet18
    from model.densenet import DenseNet
    from model.densenet3 import CIFAR_DenseNet, Bottleneck as CBott
    g = torch.Generator().manual_seed(606)
    out = {}
    cases = {
        "resnet18": (lambda: resnet18(), (3, 3, 64, 64)),
        "CIFAR_ResNet18": (lambda: CIFAR_ResNet18(), (3, 3, 32, 32)),
        # the reference factories densenet121 / CIFAR_DenseNet121 raise TypeError:
        # construct the classes directly with the intended arguments
        "densenet121": (lambda: DenseNet(32, (6, 12, 24, 16), 64), (2, 3, 64, 64)),
        "CIFAR_DenseNet121": (lambda: CIFAR_DenseNet(CBott, [6, 12, 24, 16], growth_rate=32), (2, 3, 32, 32)),
    }
    for name, (ctor, shape) in cases.items():
        net = fill_procedural(ctor())
        x = torch.randn(*shape, generator=g)
        with torch.no_grad():
            net.eval()
            out[f"{name}_x"] = x
            out[f"{name}_eval"] = net(x)
            net.train()
            out[f"{name}_train"] = net(x)          # batch statistics; running stats get updated
            out[f"{name}_nkeys"] = np.array(len(net.state_dict()))
            bn_name = [k for k in net.state_dict() if k.endswith("running_mean")][0]
            out[f"{name}_rm_name"] = np.array(bn_name)
            out[f"{name}_rm_after"] = net.state_dict()[bn_name].clone()
    npz("g6_backbones.npz", **out)


def g6b():
    """G6b: WELL-CONDITIONED train-mode outputs of the reference backbones -- 16 images, so that every
    BatchNorm sees at least 64 samples per channel (G6's 2-3 images end on 8-12 samples: three of its four train-mode
    outputs are ill-conditioned and only pin the device loosely).  Procedural weights (fill_procedural_hash: G6's sine-wave weights leave
    near-constant channels, ill-conditioned whatever the batch) AND procedural inputs (tests/procedural.py): only the outputs and one updated running mean are stored."""
    sys.path.insert(0, os.path.dirname(OUT))
    from procedural import fill_procedural_hash as fill_procedural, procedural_input
    from model.resnet import resnet18, CIFAR_ResNet18
    from model.densenet import DenseNet
    out = {}
    cases = {
        "resnet18": (lambda: resnet18(), (16, 3, 64, 64)),                     # last maps 2 x 2: 64 samples per channel
        "CIFAR_ResNet18": (lambda: CIFAR_ResNet18(), (16, 3, 32, 32)),
        "densenet121": (lambda: DenseNet(32, (6, 12, 24, 16), 64), (16, 3, 64, 64)),
    }
    for key, (name, (ctor, shape)) in enumerate(cases.items()):
        net = fill_procedural(ctor()).train()
        x = procedural_input(*shape, key=key)
        with torch.no_grad():
            out[f"{name}_shape"] = np.array(shape)
            out[f"{name}_key"] = np.array(key)
            out[f"{name}_train"] = net(x)
            bn_name = [k for k in net.state_dict() if k.endswith("running_mean")][-1]
            out[f"{name}_rm_name"] = np.array(bn_name)
            out[f"{name}_rm_after"] = net.state_dict()[bn_name].clone()
            out[f"{name}_train_f64"] = fill_procedural(ctor()).double().train()(x.double())   # how far fp32 itself is from exact
    npz("g6b_backbones_train.npz", **out)


def g9():
    """A11 projection (nw.py:74-79) and the CLIP kernel's doubly registered parameter (nw.py:82,85): state_dict
    key lists and shapes of the reference's NWNet, and what forward() returns with those weights."""
    C, shape, n = 10, (3, 4, 4), 120
    ds = _FakeDS(n, C, shape, seed=909)
    out = {"C": C, "ds_data": ds.data, "ds_targets": np.array(ds.targets)}
    g = torch.Generator().manual_seed(910)
    xq = torch.randn(6, *shape, generator=g)
    yq = torch.randint(0, C, (6,), generator=g)
    sx = torch.randn(30, *shape, generator=g)
    sy = torch.arange(30) % C
    out.update({"xq": xq, "yq": yq, "sx": sx, "sy": sy})
    for tag, kw in (("proj", dict(feat_dim=16, proj_dim=8)),
                    ("clip", dict(kernel_type="clip")),
                    ("projclip", dict(feat_dim=16, proj_dim=8, kernel_type="clip"))):
        torch.manual_seed(3)
        feat = nn.Sequential(nn.Flatten(), nn.Linear(48, 16))
        net = NWNet(feat, C, support_dataset=ds, n_shot=2, n_shot_full=5, device="cpu", **kw)
        sd = net.state_dict()
        out[f"{tag}_keys"] = np.array(list(sd.keys()))
        out[f"{tag}_shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
        out[f"{tag}_param_names"] = np.array([k for k, _ in net.named_parameters()])
        for k, v in sd.items():
            out[f"{tag}_sd_{k}"] = v.detach().clone()
        net.eval()
        with torch.no_grad():
            out[f"{tag}_fwd"] = net(xq, yq, support_data=(sx, sy, None))
            net.precompute()
            out[f"{tag}_full_feat_shape"] = np.array(net.full_feat.shape)
            out[f"{tag}_pred_full"] = net.predict(xq, "full")
    npz("g9_state_dict.npz", **out)


if __name__ == "__main__":
    torch.set_num_threads(4)
    only = sys.argv[1:]
    for fn in (g1, g2, g3, g4, g5, g6, g6b, g7, g8, g9):
        if not only or fn.__name__ in only:
            fn()
