"""``support_influence`` with the reference's signature (util/metric.py:23-50), computed by one
vectorised HIP kernel (nwhead_amd/csrc/influence.hip) instead of a Python loop over queries."""
import torch

from .. import ops


def support_influence(softmaxes, qlabels, sweights, slabels):
    """
    softmaxes: (bs, num_classes) probabilities (= exp of NWHead output)
    qlabels:   (bs, num_classes) one-hot query labels
    sweights:  (bs, num_support) softmax weights
    slabels:   (num_support, num_classes) one-hot support labels shared by the batch.
               The reference's docstring shape (bs, num_support, num_classes) makes its Python
               loop broadcast to (bs, bs, num_support); that form is accepted
               here and returns the same (bs, bs, num_support) tensor.
    returns (bs, num_support)
    """
    qy = qlabels.argmax(-1)                       # metric.py:42
    if slabels.dim() == 3:
        # reference: for query b, indicator is (bs,N) from slabels.argmax(-1) -> row r uses slabels[r]
        # and every row uses query b's p and weights
        return torch.stack([_quirk_row(softmaxes, qy, sweights, slabels, b) for b in range(len(softmaxes))])
    sy = slabels.argmax(-1)                       # metric.py:43
    return ops.support_influence_idx(softmaxes, qy, sweights, sy)


def _quirk_row(softmaxes, qy, sweights, slabels, b):
    bs = len(softmaxes)
    out = []
    for r in range(bs):
        out.append(ops.support_influence_idx(softmaxes[b:b + 1], qy[b:b + 1], sweights[b:b + 1],
                                             slabels[r].argmax(-1))[0])
    return torch.stack(out)


# ---------------------------------------------------------------------------------------------
# Bookkeeping used by the training harness (nwhead_amd/train.py); same surface as the reference's
# util/metric.py:8-20 (acc), :52-73 (Metric), :75-116 (ECELoss), :118-150 (SmoothNLLLoss).
# ---------------------------------------------------------------------------------------------
def _to_numpy(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else x


def acc(pred, targets):
    """Fraction of equal entries of two label vectors (the reference calls sklearn's accuracy_score)."""
    import numpy as np
    return float(np.mean(np.asarray(_to_numpy(pred)) == np.asarray(_to_numpy(targets))))


class Metric:
    """Sample-weighted running mean: update_state(value, n) / result() / reset_state()."""

    def __init__(self):
        self.reset_state()

    def update_state(self, val, samples):
        if isinstance(val, torch.Tensor):
            val = val.detach().cpu().item()
        elif hasattr(val, "item"):
            val = val.item()
        self.tot_val += val * samples
        self.num_samples += samples

    def result(self):
        return self.tot_val / self.num_samples if self.num_samples else 0

    def reset_state(self):
        self.tot_val, self.num_samples = 0, 0


class ECELoss(torch.nn.Module):
    """Expected calibration error over `n_bins` equal-width confidence bins (lower, upper]:
    sum_bins |mean confidence - accuracy| * (fraction of samples in the bin).  Takes PROBABILITIES
    (the harness passes exp of the head's log-probabilities, like train.py:421) and returns a (1,) tensor."""

    def __init__(self, n_bins=15):
        super().__init__()
        self.n_bins = n_bins
        edges = torch.linspace(0, 1, n_bins + 1)
        self.bin_lowers, self.bin_uppers = edges[:-1], edges[1:]

    def forward(self, softmaxes, labels):
        conf, pred = softmaxes.max(dim=1)
        hit = pred.eq(labels).float()
        lo = self.bin_lowers.to(conf.device)[:, None]
        up = self.bin_uppers.to(conf.device)[:, None]
        member = ((conf[None, :] > lo) & (conf[None, :] <= up)).float()      # (bins, samples)
        count = member.sum(1)
        safe = count.clamp_min(1)
        gap = ((member @ conf) / safe - (member @ hit) / safe).abs()
        return (gap * count / max(len(conf), 1))[count > 0].sum().reshape(1)


class SmoothNLLLoss(torch.nn.Module):
    """NLL on log-probabilities against label-smoothed targets (1 - s on the label, s/(C-1) elsewhere)."""

    def __init__(self, weight=None, reduction="mean", smoothing=0.0):
        super().__init__()
        assert 0 <= smoothing < 1
        self.weight, self.reduction, self.smoothing = weight, reduction, smoothing

    def forward(self, log_preds, targets):
        C = log_preds.size(-1)
        with torch.no_grad():
            soft = torch.full_like(log_preds, self.smoothing / (C - 1))
            soft.scatter_(1, targets.unsqueeze(1), 1.0 - self.smoothing)
        if self.weight is not None:
            log_preds = log_preds * self.weight.unsqueeze(0)
        loss = -(soft * log_preds).sum(-1)
        return loss.mean() if self.reduction == "mean" else loss.sum() if self.reduction == "sum" else loss
