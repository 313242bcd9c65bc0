"""Per-class k-means on the device (nwhead/utils.py kmeans_per_class_device) against sklearn,
which is what the reference calls (utils.py:218-246)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _blobs(n_classes, per_class, k, d, seed, spread=0.15):
    r = np.random.RandomState(seed)
    feats, labs = [], []
    for c in range(n_classes):
        centres = r.randn(k, d) * 2 + 5 * c
        n_each = np.full(k, per_class // k)
        n_each[: per_class - n_each.sum()] += 1
        for j in range(k):
            feats.append(centres[j] + spread * r.randn(n_each[j], d))
            labs += [c] * int(n_each[j])
    x = np.concatenate(feats).astype(np.float32)
    y = np.array(labs)
    p = r.permutation(len(y))                                  # classes interleaved: not a sorted bank
    return torch.from_numpy(x[p]), torch.from_numpy(y[p])


def test_one_cluster_is_the_class_mean_like_sklearn():
    from nwhead_amd.nwhead.utils import compute_clusters
    x, y = _blobs(7, 33, 1, 48, 0)
    c_dev, y_dev = compute_clusters(x.cuda(), y.cuda(), 1)                     # 'auto' -> device
    c_ref, y_ref = compute_clusters(x, y, 1, backend="sklearn")
    assert c_dev.is_cuda and torch.equal(y_dev, y_ref)
    np.testing.assert_allclose(c_dev.cpu().numpy(), c_ref.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("k,d", [(2, 16), (3, 64), (5, 32)])
def test_lloyd_matches_sklearn_on_separated_blobs(k, d):
    from nwhead_amd.nwhead.utils import compute_clusters
    x, y = _blobs(6, 40 * k, k, d, k)
    c_dev, y_dev = compute_clusters(x.cuda(), y.cuda(), k, backend="device")
    c_ref, y_ref = compute_clusters(x, y, k, backend="sklearn")
    assert torch.equal(y_dev, y_ref)
    c_dev, c_ref = c_dev.cpu().view(6, k, d), c_ref.view(6, k, d)
    for c in range(6):                                         # same set of centres per class, any order
        dist = torch.cdist(c_dev[c], c_ref[c])
        assert float(dist.min(dim=1).values.max()) < 1e-3, (c, dist)
        assert len(set(dist.argmin(dim=1).tolist())) == k


def test_fixed_point_and_closest_on_overlapping_data():
    """No clean optimum here: hold the result to Lloyd's own invariants instead -- every centre is the mean of
    the points (of its class) nearest to it, and `closest` returns rows of the class."""
    from nwhead_amd.nwhead.utils import compute_clusters
    g = torch.Generator().manual_seed(3)
    x = torch.randn(600, 24, generator=g)
    y = torch.randint(0, 5, (600,), generator=g)
    k = 4
    cent, cy = compute_clusters(x.cuda(), y.cuda(), k, backend="device")
    cent = cent.cpu()
    for c in range(5):
        pts = x[y == c]
        cc = cent[cy == c]
        a = torch.cdist(pts, cc).argmin(1)
        for j in range(k):
            if (a == j).any():
                np.testing.assert_allclose(cc[j].numpy(), pts[a == j].mean(0).numpy(), rtol=1e-4, atol=1e-5)
    near, ny = compute_clusters(x.cuda(), y.cuda(), k, closest=True, backend="device")
    near = near.cpu()
    for row, c in zip(near, ny.tolist()):
        assert (x[y == c] == row).all(dim=1).any()


@pytest.mark.parametrize("k", [2, 3, 5])
def test_device_inertia_matches_sklearn(k):
    """Round 4: with features on the GPU the device k-means is compute_clusters' default for k > 1 too.  Its centroids are
    another local optimum than sklearn's seeding finds, so they are held to sklearn's QUALITY, not its values: the summed
    squared distance of every point to its nearest own-class centroid (inertia) within 5 % of sklearn's, per class within
    25 %, on overlapping data (no clean optimum) and on blobs."""
    from nwhead_amd.nwhead.utils import compute_clusters
    g = torch.Generator().manual_seed(10 + k)
    xo = torch.randn(900, 24, generator=g) * (1 + torch.rand(1, 24, generator=g))
    yo = torch.randint(0, 6, (900,), generator=g)
    for x, y in ((xo, yo), _blobs(6, 40 * k, k, 16, k)):
        c_dev, y_dev = compute_clusters(x.cuda(), y.cuda(), k)                 # 'auto': the device
        c_ref, y_ref = compute_clusters(x, y, k, backend="sklearn")
        assert c_dev.is_cuda and torch.equal(y_dev, y_ref)
        c_dev = c_dev.cpu()
        tot_d = tot_r = 0.0
        for c in y.unique().tolist():
            pts = x[y == c].double()
            i_d = float(torch.cdist(pts, c_dev[y_dev == c].double()).min(1).values.pow(2).sum())
            i_r = float(torch.cdist(pts, c_ref[y_ref == c].double()).min(1).values.pow(2).sum())
            assert i_d <= 1.25 * i_r + 1e-6, (c, i_d, i_r)
            tot_d, tot_r = tot_d + i_d, tot_r + i_r
        assert tot_d <= 1.05 * tot_r + 1e-6, (tot_d, tot_r)
