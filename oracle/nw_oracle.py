"""CPU oracle for the NW head hot path -- TEST INFRASTRUCTURE, not product code.

Restates, op for op, what the reference computes on its PyTorch CPU path
(reference = alanqrwang/nwhead @ 2024_08_07; citations are ``file:line``
relative to the reference root).  The arithmetic itself lives in a
third-party dependency of the reference (PyTorch ATen: ``torch.cdist``,
``softmax``, ``bmm``, ``one_hot``, ``log``; the reference pins torch 1.10.1
only in prose, README.md:175-179).  Parity is pinned by the fixtures under
``tests/golden/`` which were produced by importing the reference itself in
the build container (``tests/golden/make_goldens.py``); this module is
checked against every one of them in ``tests/test_oracle_goldens.py``.

Two flavours are provided:

* ``*_f32``  -- the reference's own op sequence in fp32 torch-CPU ops.  This is
  what ``bench.py`` times as ``cpu_baseline`` (kind "port").
* ``*_f64``  -- the same mathematics evaluated in float64 with the direct
  difference form, used as the "true value" the HIP kernels are graded
  against next to the fp32 reference rounding.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

SCORE_KINDS = ("euclidean", "hypersphere_euclidean", "cosine", "dotproduct", "clip")
LOG_EPS = 1e-12                      # nwhead/nw.py:289
CLIP_LOGIT_SCALE_INIT = math.log(1 / 0.07)  # nwhead/kernel.py:38


# --------------------------------------------------------------------------
# L1: similarity kernels, nwhead/kernel.py:13-44
# --------------------------------------------------------------------------
def scores_f32(x: torch.Tensor, y: torch.Tensor, kind: str = "euclidean",
               logit_scale: float | torch.Tensor = CLIP_LOGIT_SCALE_INIT) -> torch.Tensor:
    """x:(B,nq,d) y:(B,ns,d) -> (B,nq,ns), or 2-D (nq,d),(ns,d) -> (nq,ns).

    euclidean             nwhead/kernel.py:13-15   -cdist
    hypersphere_euclidean nwhead/kernel.py:17-21   normalize, -cdist
    cosine                nwhead/kernel.py:23-28   normalize, bmm
    dotproduct            nwhead/kernel.py:30-33   bmm
    clip                  nwhead/kernel.py:35-44   normalize, exp(logit_scale)*bmm
    """
    if kind == "euclidean":
        return -torch.cdist(x, y)
    if kind == "hypersphere_euclidean":
        return -torch.cdist(F.normalize(x, dim=-1), F.normalize(y, dim=-1))
    if kind == "dotproduct":
        return torch.matmul(x, y.transpose(-2, -1))
    if kind == "cosine":
        return torch.matmul(F.normalize(x, dim=-1), F.normalize(y, dim=-1).transpose(-2, -1))
    if kind == "clip":
        ls = logit_scale if torch.is_tensor(logit_scale) else torch.tensor(float(logit_scale), dtype=x.dtype)
        return ls.exp() * torch.matmul(F.normalize(x, dim=-1), F.normalize(y, dim=-1).transpose(-2, -1))
    raise NotImplementedError(kind)      # nwhead/kernel.py:95-96


# --------------------------------------------------------------------------
# L2: NWHead.forward, nwhead/nw.py:266-289
# --------------------------------------------------------------------------
def nw_head_f32(x, sx, sy, n_classes, kind="euclidean", logit_scale=CLIP_LOGIT_SCALE_INIT,
                return_weights=False):
    """x:(B,d); sx:(N,d)|(B,N,d); sy:(N,)|(B,N) int64 -> (B,C) log-probs."""
    b = len(x)
    onehot = F.one_hot(sy, n_classes).float()                 # nw.py:276
    if sx.dim() == x.dim():                                   # nw.py:277-279
        sx = sx[None].expand(b, *sx.shape)
        onehot = onehot[None].expand(b, *onehot.shape)
    s = scores_f32(x.unsqueeze(1), sx, kind, logit_scale)     # nw.py:281-283
    w = F.softmax(s, dim=-1)                                  # nw.py:285
    out = torch.bmm(w, onehot).squeeze(1)                     # nw.py:287-288
    out = torch.log(out + LOG_EPS)                            # nw.py:289
    if return_weights:
        return out, w.squeeze(1)
    return out


def scores_f64(x, sx, kind="euclidean", logit_scale=CLIP_LOGIT_SCALE_INIT):
    """float64 scores, direct-difference form. x:(B,d), sx:(N,d)|(B,N,d) -> (B,N)."""
    x = x.double()
    sx = sx.double()
    if sx.dim() == 2:
        sx = sx[None].expand(len(x), *sx.shape)
    xq = x[:, None, :]
    if kind in ("hypersphere_euclidean", "cosine", "clip"):
        xq = xq / xq.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        sx = sx / sx.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    if kind in ("euclidean", "hypersphere_euclidean"):
        return -(xq - sx).pow(2).sum(-1).sqrt()
    dot = (xq * sx).sum(-1)
    if kind == "clip":
        ls = float(logit_scale) if not torch.is_tensor(logit_scale) else float(logit_scale.detach())
        return math.exp(ls) * dot
    if kind in ("cosine", "dotproduct"):
        return dot
    raise NotImplementedError(kind)


def nw_head_f64(x, sx, sy, n_classes, kind="euclidean", logit_scale=CLIP_LOGIT_SCALE_INIT,
                return_weights=False):
    s = scores_f64(x, sx, kind, logit_scale)
    w = torch.softmax(s, dim=-1)
    onehot = F.one_hot(sy, n_classes).double()
    if onehot.dim() == 2:
        out = w @ onehot
    else:
        out = torch.einsum("bn,bnc->bc", w, onehot)
    out = torch.log(out + LOG_EPS)
    if return_weights:
        return out, w
    return out


# --------------------------------------------------------------------------
# sharded partials (new capability). m/den/num per shard and merge.
# --------------------------------------------------------------------------
def nw_partials_f64(x, sx, sy, n_classes, kind="euclidean", logit_scale=CLIP_LOGIT_SCALE_INIT):
    """Per-shard (m, den, num): m=max_j s, den=sum_j e^(s-m), num[c]=sum_{j:sy=c} e^(s-m)."""
    s = scores_f64(x, sx, kind, logit_scale)
    m = s.max(dim=-1).values
    e = torch.exp(s - m[:, None])
    den = e.sum(-1)
    num = e @ F.one_hot(sy, n_classes).double()
    return m, den, num


def nw_merge_f64(ms, dens, nums):
    """Merge lists of shard partials -> (B,C) log-probs (SURVEY 8e formula)."""
    M = torch.stack(ms).max(dim=0).values
    den = sum(d * torch.exp(m - M) for m, d in zip(ms, dens))
    num = sum(n * torch.exp(m - M)[:, None] for m, n in zip(ms, nums))
    return torch.log(num / den[:, None] + LOG_EPS)


# --------------------------------------------------------------------------
# autograd of NWHead (implicit in the reference: loss.backward(), train.py:414)
# closed form, euclidean kernel, shared 2-D support.
# --------------------------------------------------------------------------
def nw_head_bwd_f64(x, sx, sy, n_classes, gout):
    """Returns (grad_x:(B,d), grad_sx:(N,d)) in float64 for the euclidean kernel."""
    x = x.double(); sx = sx.double(); g = gout.double()
    diff = x[:, None, :] - sx[None, :, :]
    D = diff.pow(2).sum(-1).sqrt()                       # (B,N)
    W = torch.softmax(-D, dim=-1)
    Y = F.one_hot(sy, n_classes).double()
    P = W @ Y
    dP = g / (P + LOG_EPS)
    dW = dP[:, sy]                                       # (B,N)
    dS = W * (dW - (W * dW).sum(-1, keepdim=True))
    dD = -dS
    R = torch.where(D == 0, torch.zeros_like(D), dD / D)
    gx = R.sum(1, keepdim=True) * x - R @ sx
    gs = R.sum(0)[:, None] * sx - R.t() @ x
    return gx, gs


# --------------------------------------------------------------------------
# util/metric.py:23-50 support_influence
# --------------------------------------------------------------------------
def support_influence_f32(softmaxes, qlabels, sweights, slabels):
    """softmaxes:(B,C) qlabels:(B,C) one-hot, sweights:(B,N), slabels:(N,C) one-hot -> (B,N).

    Follows util/metric.py:35-50: per query b, p = softmax[b, qy_b];
    ind_j = [argmax(slabels_j) == qy_b]; log((p - p*w)/(p - w*ind)).
    """
    rows = []
    scat = slabels.argmax(-1)                            # metric.py:43
    for b in range(len(softmaxes)):                      # metric.py:37
        qcat = int(qlabels[b].argmax(-1))                # metric.py:42
        p = softmaxes[b][qcat]                           # metric.py:45
        ind = (scat == qcat).long()                      # metric.py:46
        w = sweights[b]
        rows.append(torch.log((p - p * w) / (p - w * ind))[None])   # metric.py:47
    return torch.cat(rows, dim=0)


def as_numpy(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
