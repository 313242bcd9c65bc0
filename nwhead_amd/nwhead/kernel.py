"""Similarity modules with the reference's names and call convention (nwhead/kernel.py:13-97).

Each module is a thin handle: ``kind`` selects the score function compiled into the HIP kernels
(nwhead_amd/csrc/scores.hip).  Calling a module computes scores only (no autograd; gradients flow
through ``NWHead``, which fuses score -> softmax -> label aggregation in one autograd node).

    forward(x:(B,nq,d), y:(B,ns,d)) -> (B,nq,ns)      reference convention, kernel.py:6-11
    forward(x:(nq,d),   y:(ns,d))   -> (nq,ns)        2-D use, nw.py:248 (cdist kinds only there)
"""
import math

import torch
import torch.nn as nn

from .. import ops


class _ScoreModule(nn.Module):
    kind = None

    def _logit_scale(self):
        return None

    def forward(self, x, y):
        ls = self._logit_scale()
        if x.dim() == 2 and y.dim() == 2:
            return ops.nw_scores(x, y, self.kind, ls)
        if x.dim() != 3 or y.dim() != 3:
            raise ValueError("expected (B,nq,d) and (B,ns,d)")
        if x.shape[1] == 1:
            # a stride-0 batch (nw.py:278 expands one shared support) is passed once, not B times
            sup = y[0] if (y.stride(0) == 0 or y.shape[0] == 1) else y
            return ops.nw_scores(x[:, 0], sup, self.kind, ls).unsqueeze(1)
        return torch.stack([ops.nw_scores(x[b], y[b], self.kind, ls) for b in range(x.shape[0])])


class EuclideanDistance(_ScoreModule):          # kernel.py:13-15
    kind = "euclidean"


class HypersphereEuclideanDistance(_ScoreModule):  # kernel.py:17-21
    kind = "hypersphere_euclidean"


class CosineDistance(_ScoreModule):             # kernel.py:23-28
    kind = "cosine"


class DotProduct(_ScoreModule):                 # kernel.py:30-33
    kind = "dotproduct"


class Clip(_ScoreModule):                       # kernel.py:35-44
    kind = "clip"

    def __init__(self):
        super().__init__()
        self.logit_scale = nn.Parameter(torch.ones([]) * math.log(1 / 0.07))

    def _logit_scale(self):
        return self.logit_scale


_KERNELS = {
    "euclidean": EuclideanDistance,
    "hypersphere_euclidean": HypersphereEuclideanDistance,
    "cosine": CosineDistance,
    "dotproduct": DotProduct,
    "clip": Clip,
}


def get_kernel(kernel_type):
    """kernel.py:80-97: unknown names raise NotImplementedError."""
    try:
        return _KERNELS[kernel_type]()
    except KeyError:
        raise NotImplementedError
