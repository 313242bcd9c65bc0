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
