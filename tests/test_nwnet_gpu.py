"""NWNet API parity on the MI355X against fixture G5 (reference NWNet with a tiny featurizer)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import T, load_golden

pytestmark = pytest.mark.gpu


class _DS(torch.utils.data.Dataset):
    def __init__(self, data, targets, C):
        self.data, self.targets, self.num_classes = data, list(targets), C

    def __len__(self):
        return len(self.targets)

    def __getitem__(self, i):
        return self.data[i], self.targets[i]


@pytest.fixture(scope="module")
def net_and_g():
    from nwhead_amd.nwhead.nw import NWNet
    g = load_golden("g5_nwnet_plumbing.npz")
    C = int(g["C"])
    ds = _DS(T(g["ds_data"]), g["ds_targets"].tolist(), C)
    feat = nn.Sequential(nn.Flatten(), nn.Linear(48, 16))
    with torch.no_grad():
        feat[1].weight.copy_(T(g["w"]))
        feat[1].bias.copy_(T(g["b"]))
    # cluster_backend="sklearn": G5's cluster-mode outputs are those of the reference's sklearn call (utils.py:230) with two
    # clusters per class, an optimum that depends on sklearn's seeding; the default on the device is the device k-means
    # (another optimum of the same quality: test_kmeans_gpu.py::test_device_inertia_matches_sklearn)
    net = NWNet(feat, C, support_dataset=ds, feat_dim=16, n_shot=2, n_way=6, n_shot_full=7,
                n_shot_cluster=2, n_neighbors=3, device="cuda:0", cluster_backend="sklearn").to("cuda:0")
    net.eval()
    np.random.seed(1234)
    net.precompute()
    return net, g


def close(a, b, rtol=1e-5, atol=3e-5):
    np.testing.assert_allclose(a.detach().cpu().numpy(), b, rtol=rtol, atol=atol)


def test_bank_order_and_state_dict(net_and_g):
    net, g = net_and_g
    assert net.full_feat.is_cuda                      # bank stays device-resident
    np.testing.assert_array_equal(net.full_y.cpu().numpy(), g["full_y"])   # class-sorted, balanced
    close(net.full_feat, g["full_feat"], rtol=1e-5, atol=1e-5)
    assert set(net.state_dict().keys()) == {"featurizer.1.weight", "featurizer.1.bias"}
    np.testing.assert_array_equal(net.support_eval.cluster_y.cpu().numpy(), g["cluster_y"])


@pytest.mark.parametrize("mode", ["full", "cluster", "knn", "hnsw", "ensemble"])
def test_predict_modes(net_and_g, mode):
    net, g = net_and_g
    xq = T(g["xq"]).cuda()
    with torch.no_grad():
        out = net.predict(xq, mode)
    close(out, g[f"pred_{mode}"], rtol=1e-4, atol=1e-4 if mode == "cluster" else 3e-5)


def test_random_mode_and_sampler_draws(net_and_g):
    net, g = net_and_g
    xq, yq = T(g["xq"]).cuda(), T(g["yq"]).cuda()
    with torch.no_grad():
        np.random.seed(77)
        close(net.predict(xq, "random"), g["pred_random"])
        np.random.seed(77)
        _, ry, _ = net.support_eval.random_iter.next()
        np.testing.assert_array_equal(ry.cpu().numpy(), g["random_sy"])
        np.random.seed(99)
        sx, sy, sm = net.support_train.get_support(yq)
        np.testing.assert_array_equal(sy.numpy(), g["train_sy"])        # incl. duplicate-class n_way case
        np.testing.assert_array_equal(sx.numpy(), g["train_sx"])
        close(net(xq, yq, support_data=(sx, sy, None)), g["fwd_support_data"])
        np.random.seed(99)
        close(net(xq, yq), g["fwd_sampled"])
    nb = net.get_neighbors(xq).cpu().numpy()
    # same nearest neighbours (ties aside): compare the top-5 columns
    np.testing.assert_array_equal(nb[:, :5], g["neighbors"][:, :5])


def test_errors(net_and_g):
    net, _ = net_and_g
    from nwhead_amd.nwhead.kernel import get_kernel
    with pytest.raises(NotImplementedError):
        get_kernel("relationnet")
    with pytest.raises(NotImplementedError):
        net.support_eval.get_support("nope")


@pytest.mark.parametrize("arch", ["CIFAR_ResNet18", "resnet18"])
def test_bn_folding_through_nwnet(arch):
    """enable_bn_folding(): precompute() / predict() run the folded copy in eval mode, train() drops it, the
    state_dict is unchanged, and the predictions agree with the unfolded network."""
    from nwhead_amd.data import SyntheticImages
    from nwhead_amd.model import load_model
    from nwhead_amd.nwhead.nw import NWNet
    from tests.procedural import fill_procedural
    ds = SyntheticImages(6, 5, 32, seed=2)
    feat = load_model(arch)
    fill_procedural(feat)
    net = NWNet(feat, 5, support_dataset=ds, n_shot_full=6, device="cuda:0").to("cuda:0").eval()
    x = torch.stack([ds[i][0] for i in range(0, 30, 3)]).to("cuda:0")
    keys = list(net.state_dict().keys())
    with torch.no_grad():
        net.precompute()
        ref = net.predict(x, "full")
        net.enable_bn_folding()
        net.precompute()
        assert net._folded is not None
        # the copies are kept in channels_last (a copy with NCHW HIP kernels inside -- ScaleShiftReLU, Conv1x1Fused -- would not be)
        assert hasattr(net._folded, "inner")
        if arch == "resnet18":
            assert not any(isinstance(m, nn.BatchNorm2d) for m in net._folded.modules())
        out = net.predict(x, "full")
        if arch != "resnet18":   # round 4: CIFAR_ResNet._forward_nhwc_infer folds for itself from the plain modules (as the DenseNets' path does)
            assert getattr(net._folded.inner, "_nw_infer_plan", None) is not None
    assert torch.allclose(out, ref, rtol=1e-3, atol=1e-3), (out - ref).abs().max()
    assert list(net.state_dict().keys()) == keys
    net.train()
    assert net._folded is None
    net.eval()


@pytest.mark.parametrize("shape,prefix", [((3, 40, 14, 14), 24), ((2, 17, 7, 7), 17), ((1, 8, 5, 3), 5), ((4, 64, 56, 56), 64)])
@pytest.mark.parametrize("relu", [True, False])
def test_scale_shift_relu_kernel(shape, prefix, relu):
    """nw_scale_shift_relu_f32 (eval BatchNorm + ReLU in one pass) on a channel prefix of a wider slab,
    float4 and scalar planes, against the two torch ops it replaces."""
    from nwhead_amd import ops
    g = torch.Generator().manual_seed(sum(shape))
    slab = torch.randn(*shape, generator=g).cuda()
    a, b = torch.randn(prefix, generator=g).cuda(), torch.randn(prefix, generator=g).cuda()
    x = slab[:, :prefix]
    got = ops.scale_shift_relu(x, a, b, relu)
    ref = x * a.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    ref = torch.relu(ref) if relu else ref
    assert got.is_contiguous() and got.shape == x.shape
    torch.testing.assert_close(got, ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("arch,size", [("densenet121", 96), ("CIFAR_DenseNet121", 32), ("CIFAR_ResNet18", 32)])
def test_preactivation_folded_copy_on_the_device(arch, size):
    """fold_batchnorm on the pre-activation nets: inner conv+BN pairs folded, the BatchNorm -> ReLU pairs through
    the HIP kernel; same features as the eval-mode network."""
    from nwhead_amd.model import fold_batchnorm, load_model
    from tests.procedural import fill_procedural
    net = load_model(arch)
    fill_procedural(net)
    net = net.cuda().train()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        net(torch.randn(4, 3, size, size, generator=g).cuda())    # running statistics off their init
        net.eval()
        folded = fold_batchnorm(net)
        x = torch.randn(3, 3, size, size, generator=g).cuda()
        a, b = net(x), folded(x)
    if arch in ("densenet121", "CIFAR_ResNet18"):     # round 4: served by _forward_nhwc_infer, which folds for itself from the plain modules
        assert getattr(folded, "_nw_infer_plan", None) is not None
    else:
        assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in folded.modules())
    torch.testing.assert_close(b, a, rtol=1e-4, atol=1e-5 * a.abs().max().item())


def test_knn_mode_with_per_query_neighbours(net_and_g):
    """NWNet(knn_per_query=True) (not in the reference, whose knn mode concatenates ALL queries' neighbours into
    one shared support): every query attends to ITS OWN k nearest bank rows -- (B, k, d) supports through the head's
    per-query path -- against torch on the same bank."""
    from nwhead_amd.nwhead.nw import NWNet
    net0, g = net_and_g
    C = int(g["C"])
    ds = _DS(T(g["ds_data"]), g["ds_targets"].tolist(), C)
    feat = nn.Sequential(nn.Flatten(), nn.Linear(48, 16))
    feat.load_state_dict(net0.featurizer.state_dict())
    net = NWNet(feat, C, support_dataset=ds, feat_dim=16, n_shot_full=7, n_neighbors=5, device="cuda:0",
                knn_per_query=True).to("cuda:0").eval()
    net.precompute()
    xq = T(g["xq"]).cuda()
    with torch.no_grad():
        out = net.predict(xq, "knn")
        q = net.featurizer(xq)
        dist = torch.cdist(q.double(), net.full_feat.double())
        idx = dist.argsort(dim=1, stable=True)[:, :5]
        dk = torch.gather(dist, 1, idx)
        w = torch.softmax(-dk, -1)
        yk = net.full_y[idx]
        want = torch.log(torch.stack([(w * (yk == c)).sum(-1) for c in range(C)], -1) + 1e-12)
    assert net.support_eval.knn(q)[0].shape == (xq.shape[0], 5, 16)
    close(out, want.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
