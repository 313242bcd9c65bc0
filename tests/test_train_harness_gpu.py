"""End to end on the MI355X: the harness trains a small NWNet on procedural images through the HIP head
(forward + backward), evaluates the three support modes after precompute(), checkpoints and resumes."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_harness_trains_checkpoints_and_resumes(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from nwhead_amd.train import Trainer, make_parser
    argv = ["--dataset", "synthetic", "--arch", "tiny", "--models_dir", str(tmp_path), "--batch_size", "40",
            "--lr", "0.05", "--num_epochs", "4", "--log_interval", "2", "--seed", "1", "--n_shot", "2",
            "--synthetic_classes", "6", "--synthetic_per_class", "20", "--synthetic_size", "32",
            "--synthetic_noise", "2.5", "--scheduler_milestones", "3"]
    tr = Trainer(make_parser().parse_args(argv))
    hist = tr.fit()
    assert [h["epoch"] for h in hist] == [1, 2, 3, 4]
    for key in ("loss:train", "acc:train", "loss:val:random", "acc:val:full", "ece:val:cluster"):
        assert key in hist[0]
    assert hist[-1]["loss:train"] < hist[0]["loss:train"]                    # it learns
    assert hist[-1]["acc:val:full"] > 100.0 / 6 + 20                          # far above chance (16.7 %)
    assert abs(hist[2]["lr"] - 0.005) < 1e-9                                  # MultiStepLR stepped at 3
    ck = os.path.join(tr.ckpt_dir, "model.0004.h5")
    assert os.path.exists(ck) and os.path.exists(os.path.join(tr.ckpt_dir, "model.0002.h5"))
    # model.best.h5 is written when a checkpointing epoch (2, 4) set a new best 'full' accuracy (train.py:306-311)
    best, expect_best = 0.0, False
    for h in hist:
        if h["acc:val:full"] > best and h["epoch"] % 2 == 0:
            expect_best = True
        best = max(best, h["acc:val:full"])
    assert os.path.exists(os.path.join(tr.ckpt_dir, "model.best.h5")) == expect_best
    # resume: picks up after epoch 4 with the saved optimizer / scheduler state
    tr2 = Trainer(make_parser().parse_args(argv[:-2] + ["--scheduler_milestones", "3", "--resume"]
                                           + ["--num_epochs", "5"]))
    assert tr2.start_epoch == 5 and tr2.best_acc1 == tr.best_acc1
    for a, b in zip(tr.network.state_dict().values(), tr2.network.state_dict().values()):
        assert torch.equal(a.cpu(), b.cpu())
    hist2 = tr2.fit()
    assert [h["epoch"] for h in hist2] == [5] and abs(hist2[0]["lr"] - 0.005) < 1e-9
