"""Host-side pieces of the training harness (nwhead_amd/train.py, util/, data.py): no GPU needed."""
import os
import pickle

import numpy as np
import torch
import torch.nn as nn

from nwhead_amd.data import CIFARFromDisk, SyntheticImages
from nwhead_amd.util.metric import ECELoss, Metric, SmoothNLLLoss, acc
from nwhead_amd.util.utils import latest_checkpoint, load_checkpoint, save_checkpoint


def test_metric_running_mean():
    m = Metric()
    assert m.result() == 0
    m.update_state(torch.tensor(2.0), 3)
    m.update_state(np.float32(4.0), 1)
    assert abs(m.result() - 2.5) < 1e-12
    m.reset_state()
    assert m.result() == 0 and m.num_samples == 0


def test_acc():
    assert acc(torch.tensor([1, 2, 3, 3]), torch.tensor([1, 0, 3, 2])) == 0.5


def _ece_loop(probs, labels, n_bins=15):
    """the definition, bin by bin (util/metric.py:99-116 semantics: bins (lower, upper])"""
    conf, pred = probs.max(1)
    hit = pred.eq(labels).float()
    edges = torch.linspace(0, 1, n_bins + 1)
    ece = 0.0
    for lo, up in zip(edges[:-1], edges[1:]):
        m = (conf > lo.item()) & (conf <= up.item())
        if m.any():
            ece += (conf[m].mean() - hit[m].mean()).abs().item() * m.float().mean().item()
    return ece


def test_ece_matches_definition():
    g = torch.Generator().manual_seed(0)
    probs = torch.softmax(torch.randn(500, 7, generator=g) * 2, -1)
    labels = torch.randint(0, 7, (500,), generator=g)
    out = ECELoss()(probs, labels)
    assert out.shape == (1,)
    assert abs(out.item() - _ece_loop(probs, labels)) < 1e-6
    sure = torch.eye(4)[torch.tensor([0, 1, 2, 3])]
    assert ECELoss()(sure, torch.tensor([0, 1, 2, 3])).item() < 1e-7        # confident and right


def test_smooth_nll():
    lp = torch.log_softmax(torch.randn(5, 4), -1)
    y = torch.tensor([0, 1, 2, 3, 0])
    assert torch.allclose(SmoothNLLLoss()(lp, y), nn.NLLLoss()(lp, y))
    s = 0.1
    want = -(lp.gather(1, y[:, None]).squeeze(1) * (1 - s) + (lp.sum(1) - lp.gather(1, y[:, None]).squeeze(1)) * s / 3)
    assert torch.allclose(SmoothNLLLoss(smoothing=s)(lp, y), want.mean())


def test_checkpoint_roundtrip_and_latest(tmp_path):
    net = nn.Sequential(nn.Linear(3, 4), nn.BatchNorm1d(4))
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, nesterov=True)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[2], gamma=0.1)
    net(torch.randn(8, 3)).sum().backward()
    opt.step()
    sched.step()
    assert latest_checkpoint(str(tmp_path)) is None
    save_checkpoint(3, net, opt, str(tmp_path), sched, is_best=True, extra={"best_acc1": 42.0})
    save_checkpoint(12, net, opt, str(tmp_path), sched)
    assert os.path.exists(tmp_path / "model.0003.h5") and os.path.exists(tmp_path / "model.best.h5")
    assert latest_checkpoint(str(tmp_path)).endswith("model.0012.h5")
    net2 = nn.Sequential(nn.Linear(3, 4), nn.BatchNorm1d(4))
    opt2 = torch.optim.SGD(net2.parameters(), lr=0.5, momentum=0.9, nesterov=True)
    sched2 = torch.optim.lr_scheduler.MultiStepLR(opt2, milestones=[2], gamma=0.1)
    ck = load_checkpoint(net2, str(tmp_path / "model.0003.h5"), opt2, sched2, verbose=False)
    assert ck["epoch"] == 3 and ck["best_acc1"] == 42.0
    for a, b in zip(net.state_dict().values(), net2.state_dict().values()):
        assert torch.equal(a, b)
    assert opt2.state_dict()["param_groups"][0]["lr"] == opt.state_dict()["param_groups"][0]["lr"]
    assert sched2.last_epoch == sched.last_epoch


def test_synthetic_images_deterministic_and_class_sorted():
    a, b = SyntheticImages(4, 5, 16, seed=3), SyntheticImages(4, 5, 16, seed=3)
    assert len(a) == 20 and a.num_classes == 5 and a.targets == sorted(a.targets)
    x0, y0 = a[7]
    x1, y1 = b[7]
    assert y0 == y1 == 1 and torch.equal(x0, x1) and x0.shape == (3, 16, 16)
    assert not torch.equal(a[7][0], a[6][0])                    # same class, different noise
    # class structure is there: a sample is closer to its own prototype than to another class's
    assert (x0 - a.protos[1]).norm() < (x0 - a.protos[3]).norm()


def test_cifar_from_disk(tmp_path):
    base = tmp_path / "cifar-10-batches-py"
    base.mkdir()
    rng = np.random.RandomState(0)
    for i in range(1, 6):
        with open(base / f"data_batch_{i}", "wb") as fh:
            pickle.dump({b"data": rng.randint(0, 256, (4, 3072), dtype=np.uint8), b"labels": [i % 10] * 4}, fh)
    with open(base / "test_batch", "wb") as fh:
        pickle.dump({b"data": rng.randint(0, 256, (3, 3072), dtype=np.uint8), b"labels": [1, 2, 3]}, fh)
    tr, te = CIFARFromDisk(str(tmp_path), True), CIFARFromDisk(str(tmp_path), False)
    assert len(tr) == 20 and len(te) == 3 and te.targets == [1, 2, 3] and tr.num_classes == 10
    x, y = te[0]
    assert x.shape == (3, 32, 32) and y == 1
    raw = te.data[0].float() / 255
    assert torch.allclose(x, (raw - te.mean) / te.std)          # test split: normalisation only
    np.random.seed(0)
    xa, _ = tr[0]
    assert xa.shape == (3, 32, 32)
