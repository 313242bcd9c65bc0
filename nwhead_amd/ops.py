"""Tensor-level entry points over the C ABI: forward/backward of the NW head, scores, sharded
partials + merge, support influence.  PyTorch is plumbing only (device memory, current stream,
autograd bookkeeping); every arithmetic step runs in libnwhead_hip.so.
"""
from __future__ import annotations

import ctypes
import os

import torch
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import SCORE_KINDS, NWHipError


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t: torch.Tensor):
    """The current HIP stream of t's device as the integer handle the C ABI takes (torch.cuda.current_stream builds
    a Python Stream object per call: ~4 us; the raw getter is what it wraps)."""
    if _raw_stream is not None:
        idx = t.device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(t.device).cuda_stream


class _OnDevice:
    """`with torch.cuda.device(dev)` only when `dev` is not already the current device (the context manager costs
    several microseconds per call; the C ABI launches on the CURRENT device)."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        idx = dev.index
        self.ctx = None if (idx is None or idx == torch.cuda.current_device()) else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)


def _sig(t):
    """(storage address, shape, in-place version) of a tensor: what 'the very tensor, unmodified' means for the cached
    banks and run tables.  Inference tensors (torch.inference_mode) carry no version counter: None stands in."""
    return (t.data_ptr(), tuple(t.shape), _ver(t))


def _ver(t):
    """In-place version of a tensor, None for inference tensors (they have no counter and cannot be written in place
    outside inference mode; a replaced one shows up in its address)."""
    return None if t.is_inference() else t._version


_WS_BYTES = {}


def _fwd_ws_bytes(lib, B, N, d, C):
    key = (B, N, d, C)
    v = _WS_BYTES.get(key)
    if v is None:
        if len(_WS_BYTES) > 4096:
            _WS_BYTES.clear()
        v = _WS_BYTES[key] = lib.nw_fwd_workspace_bytes(B, N, d, C)
    return v


def _ptr(t):
    return None if t is None else t.data_ptr()


def _need_hip(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise NWHipError(
                "nwhead_amd ops run on MI355X only: got a CPU tensor. Move the module and its "
                "inputs to the HIP device (there is no CPU fallback in the product path).")


def _f32c(t):
    if t.dtype == torch.float32 and t.is_contiguous():
        return t.detach() if t.requires_grad else t
    return t.detach().to(torch.float32).contiguous()


def _kind_id(kind):
    if isinstance(kind, int):
        return kind
    try:
        return SCORE_KINDS[kind]
    except KeyError:
        raise NotImplementedError(kind)


_WS_CACHE = {}
_OPTS = {}


def _default_opts(persistent_wgs=0):
    """The nw_fwd_opts of a call without cached run tables (kept alive here; NW_SPLIT_ALWAYS read when first needed)."""
    key = (int(persistent_wgs), _lib.force_split())
    op = _OPTS.get(key)
    if op is None:
        op = _OPTS[key] = _lib.fwd_opts(persistent_wgs=persistent_wgs)
    return C_addr(op)


def C_addr(op):
    import ctypes
    return ctypes.addressof(op)


def _workspace(nbytes, device, stream=None):
    """Scratch for one call.  Cached per (device, stream) and only ever grown: work on one stream is
    ordered, so the next call may reuse it; another stream gets its own."""
    nbytes = max(int(nbytes), 1)
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           stream if stream is not None else torch.cuda.current_stream(device).cuda_stream)
    ws = _WS_CACHE.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _WS_CACHE[key] = ws
    return ws


def nw_scores(q, s, kind="euclidean", logit_scale=None, support_cache=None):
    """q:(B,d), s:(N,d)|(B,N,d) -> (B,N) fp32 scores.  Replaces nwhead/kernel.py:13-44.
    support_cache: the SplitBank of `s` (a resident bank, e.g. the neighbour search over precompute()'s features): the
    scores then come from the split-fp16 tile kernel (the forward with its score output; 2-3x the fp32 scores kernel)."""
    _need_hip(q, s, logit_scale)
    lib = _lib.load()
    q, s = _f32c(q), _f32c(s)
    B, d = q.shape
    batched = s.dim() == 3
    N = s.shape[-2]
    out = torch.empty(B, N, dtype=torch.float32, device=q.device)
    ls = None if logit_scale is None else _f32c(logit_scale)
    if (support_cache is not None and not batched and support_cache.split is not None and support_cache.sorted_rows is None
            and support_cache.matches(s) and B > 0 and N > 0 and N % 4 == 0):
        q, s = _apply_bank_padding(q, s, support_cache)
        d = q.shape[1]
        # one class, all labels 0: the aggregation is a formality, the (B, N) score matrix is what is wanted
        zeros = getattr(support_cache, "_zero_labels", None)
        if zeros is None or zeros.numel() != N:
            zeros = support_cache._zero_labels = torch.zeros(N, dtype=torch.int64, device=q.device)
        out1 = torch.empty(B, 1, dtype=torch.float32, device=q.device)
        ws_bytes = _fwd_ws_bytes(lib, B, N, d, 1)
        ws = _workspace(ws_bytes, q.device) if ws_bytes else None
        with _OnDevice(q.device):
            _lib.check(lib.nw_fwd_f32(_ptr(q), _ptr(s), _ptr(zeros), _ptr(support_cache.norm2), _ptr(support_cache.split),
                                      _ptr(support_cache.scale), _ptr(out1), _ptr(out), None, None, _ptr(ws), ws_bytes,
                                      B, N, d, 1, _kind_id(kind), _ptr(ls), 0, 0, _default_opts(), _stream(q)), "nw_fwd_f32")
        return out
    with torch.cuda.device(q.device):
        _lib.check(lib.nw_scores_f32(_ptr(q), _ptr(s), _ptr(out), B, N, d, _kind_id(kind), _ptr(ls),
                                     int(batched), _stream(q)), "nw_scores_f32")
    return out


def nw_topk(scores, k, return_values=False):
    """scores (B,N) fp32 HIP tensor -> (B,k) int64 column indices, best score first, ties by ascending
    index: the first k columns of argsort(scores, descending=True, stable=True) (nwhead/utils.py:185-193),
    without sorting the other N-k.  k <= min(N, 1024)."""
    _need_hip(scores)
    lib = _lib.load()
    sc = _f32c(scores)
    B, N = sc.shape
    k = int(k)
    idx = torch.empty(B, k, dtype=torch.int64, device=sc.device)
    vals = torch.empty(B, k, dtype=torch.float32, device=sc.device) if return_values else None
    with torch.cuda.device(sc.device):
        _lib.check(lib.nw_topk_f32(_ptr(sc), _ptr(idx), _ptr(vals), B, N, k, _stream(sc)), "nw_topk_f32")
    return (idx, vals) if return_values else idx


def row_norm2(x):
    """Squared L2 norm of every row of a (rows,d) fp32 HIP tensor -> (rows,).  Cache this for a support
    bank that does not change between calls and pass it as ``support_norm2``."""
    _need_hip(x)
    lib = _lib.load()
    xc = _f32c(x)
    out = torch.empty(xc.shape[0], dtype=torch.float32, device=xc.device)
    with torch.cuda.device(xc.device):
        _lib.check(lib.nw_row_norm2_f32(_ptr(xc), _ptr(out), xc.shape[0], xc.shape[1], _stream(xc)), "nw_row_norm2_f32")
    return out


class SplitBank:
    """A support matrix prepared once for repeated 'full' inference: split-fp16 rows, row scales and
    squared norms (nw_split_rows_f16x2).  Pass it as ``support_cache`` to nw_head / nw_partials.
    The split format needs d % 32 == 0: a bank of another width (d >= 64) keeps a copy padded with zero columns
    (``rows``, ``pad``; no dot product or norm changes) and the callers pad the queries to match; narrower ones fall back
    to norms only (fp32 matrix cores).

    ``labels``: the (N,) labels that will be used with this bank.  The tile kernels sum softmax weights
    per RUN of equal consecutive labels, so a class-sorted bank (what precompute() builds) costs 1-2 sums
    per tile and an unsorted one a sum per row (measured 1862 vs 322 us at B=2048, N=50000).  The output
    does not depend on the order of the supports, so when unsorted labels are given the bank keeps a
    class-sorted copy (``sorted_rows`` / ``sorted_labels``, stable order) and nw_head runs on that."""

    def __init__(self, s, labels=None):
        _need_hip(s, labels)
        lib = _lib.load()
        sc = _f32c(s)
        # the tensor this bank was prepared from: identity, shape and in-place version (see matches())
        self._src = _sig(s)
        self.label_max = None
        if labels is not None and labels.numel():
            lo, hi = (int(v) for v in torch.aminmax(labels.detach()))
            if lo < 0:
                raise ValueError("support labels must be non-negative class indices (F.one_hot, nw.py:276, raises too)")
            self.label_max = hi                # nw_head refuses n_classes <= label_max, like F.one_hot
        self.pad, self.rows = 0, None
        if sc.dim() == 2 and sc.shape[1] % 32 and sc.shape[1] >= 64 and sc.shape[0] > 0:
            self.pad = (-sc.shape[1]) % 32
            sc = self.rows = torch.nn.functional.pad(sc, (0, self.pad))     # what the kernels read instead of `s`
        self.sorted_rows = self.sorted_labels = None
        if labels is not None and sc.dim() == 2 and labels.dim() == 1 and labels.numel() > 1:
            lab = labels.detach().to(torch.int64)
            if bool((lab[1:] < lab[:-1]).any()):
                perm = torch.argsort(lab, stable=True)
                self.sorted_rows, self.sorted_labels = sc[perm].contiguous(), lab[perm].contiguous()
                self._labels_ref = labels          # the label tensor this bank was sorted for (identity check only)
                sc = self.sorted_rows
        N, d = sc.shape
        self.shape = (N, d)
        self.split = self.scale = None
        if d % 32 == 0 and N > 0:
            self.split = torch.empty_like(sc)
            self.scale = torch.empty(N, dtype=torch.float32, device=sc.device)
            self.norm2 = torch.empty(N, dtype=torch.float32, device=sc.device)
            with torch.cuda.device(sc.device):
                _lib.check(lib.nw_split_rows_f16x2(_ptr(sc), _ptr(self.split), _ptr(self.scale), _ptr(self.norm2),
                                                   N, d, _stream(sc)), "nw_split_rows_f16x2")
        else:
            self.norm2 = row_norm2(sc)
        self.tables = self._tables_src = None
        self.tables_label_max = -1
        self._opts = {}
        if labels is not None and labels.dim() == 1 and self.split is not None:
            self.build_tables(self.sorted_labels if self.sorted_labels is not None else labels)

    def build_tables(self, labels):
        """Run tables of this bank under ``labels`` (nw_bank_tables_build): the forward then skips building them on
        every large launch.  They are used only for calls that pass this very label tensor, unmodified."""
        lib = _lib.load()
        lab = labels.detach()
        lab64 = lab if (lab.dtype == torch.int64 and lab.is_contiguous()) else lab.to(torch.int64).contiguous()
        N = self.shape[0]
        if lab64.dim() != 1 or lab64.numel() != N or N == 0 or not lab64.is_cuda:
            return
        lo, hi = (int(v) for v in torch.aminmax(lab64))
        if lo < 0:
            raise ValueError("support labels must be non-negative class indices (F.one_hot, nw.py:276, raises too)")
        self.tables_label_max = hi      # the tables hold every label as a real class: n_classes must exceed it (call_opts)
        nbytes = lib.nw_bank_tables_bytes(N)
        tables = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=lab64.device)
        with torch.cuda.device(lab64.device):
            # C: any bound above the labels gives the same tables (nw_head refuses labels >= n_classes)
            _lib.check(lib.nw_bank_tables_build(_ptr(lab64), N, 0x7fffffff, _ptr(tables), nbytes, _stream(lab64)),
                       "nw_bank_tables_build")
        self.tables, self._tables_src, self._opts = tables, _sig(labels), {}

    def call_opts(self, sy, n_classes, persistent_wgs=0, sy_call=None):
        """Address of the nw_fwd_opts for a forward call with labels ``sy``: names the cached run tables when ``sy`` is the
        label tensor they were built from (same storage, unmodified) -- the object stays alive in this bank.  ``sy_call``:
        the int64 tensor whose address the call passes as its labels (``sy`` itself unless it had to be converted); the
        library uses the tables only for that address and row count."""
        if self.tables is not None and _sig(sy) == self._tables_src:
            syc = sy if sy_call is None else sy_call
            if self.tables_label_max >= int(n_classes):
                raise ValueError(f"support label {self.tables_label_max} is outside [0, n_classes={int(n_classes)}) "
                                 "(the reference's F.one_hot, nw.py:276, raises)")
            key = (int(persistent_wgs), _lib.force_split(), syc.data_ptr(), syc.numel())
            op = self._opts.get(key)
            if op is None:
                if len(self._opts) > 64:
                    self._opts.clear()
                op = self._opts[key] = _lib.fwd_opts(self.tables.data_ptr(), self.tables.numel(), persistent_wgs,
                                                     syc.data_ptr(), syc.numel())
            return C_addr(op)
        return _default_opts(persistent_wgs)

    def matches(self, s):
        """True when `s` is the very tensor (storage, shape, no in-place update since) this bank was prepared from."""
        return _sig(s) == self._src


def _resolve_sorted_bank(s, sy, cache, per_position_outputs=False):
    """A SplitBank built from UNSORTED labels holds the split form of its class-sorted copy: run on that
    copy (same output: the order of the supports does not matter) when the caller passes the very label
    tensor the bank was built with and wants nothing indexed by support position; otherwise drop the cache
    (correct, slower) rather than pair split rows with labels in another order."""
    if cache is None or cache.sorted_rows is None:
        return s, sy, cache
    same = sy is cache._labels_ref or (sy.data_ptr() == cache._labels_ref.data_ptr() and sy.shape == cache._labels_ref.shape)
    if per_position_outputs or not same:
        return s, sy, None
    return cache.sorted_rows, cache.sorted_labels, cache


def _apply_bank_padding(q, s, cache):
    """A bank of a width that is not a multiple of 32 holds zero-padded rows (SplitBank.rows / .sorted_rows): pad the
    queries alike and read the bank's rows instead of the caller's tensor."""
    if cache is None or not cache.pad:
        return q, s
    q = torch.nn.functional.pad(q, (0, cache.pad))
    if s.shape[-1] != cache.shape[1]:
        s = cache.rows
    return q, s


class _NoCtx:
    """Stand-in for the autograd context on the inference path."""
    needs_input_grad = (False, False, False, False)

    @staticmethod
    def mark_non_differentiable(*a):
        pass


class _NWHeadFn(torch.autograd.Function):
    """autograd node for NWHead.forward (nwhead/nw.py:266-289)."""

    @staticmethod
    def forward(ctx, q, s, sy, logit_scale, n_classes, kind_id, want_weights, sn2, cache):
        _need_hip(q, s, sy, logit_scale, sn2)
        ssplit = sscale = None
        if cache is not None:
            sn2, ssplit, sscale = cache.norm2, cache.split, cache.scale
        lib = _lib.load()
        qc, sc = _f32c(q), _f32c(s)
        syc = sy if (sy.dtype == torch.int64 and sy.is_contiguous()) else sy.detach().to(torch.int64).contiguous()
        B, d = qc.shape
        sup_b = sc.dim() == 3
        lab_b = syc.dim() == 2
        N = sc.shape[-2]
        dev = qc.device
        need_bwd = any(ctx.needs_input_grad[:2]) or (logit_scale is not None and ctx.needs_input_grad[3])
        if need_bwd:
            _lib.sync_knobs()          # (NW_BWD_SPLIT and the other diagnostic switches may be flipped between steps)
        if (need_bwd and ssplit is None and not sup_b and N > 0
                and lib.nw_bwd_uses_split(B, N, d, n_classes, 0)):
            # a training step at a size where the backward's products run on split rows: split the supports once,
            # for this forward (fp16 matrix cores instead of fp32) and for the backward
            ssplit, sscale = torch.empty_like(sc), torch.empty(N, dtype=torch.float32, device=dev)
            sn2 = torch.empty(N, dtype=torch.float32, device=dev)
            with _OnDevice(dev):
                _lib.check(lib.nw_split_rows_f16x2(_ptr(sc), _ptr(ssplit), _ptr(sscale), _ptr(sn2), N, d, _stream(qc)),
                           "nw_split_rows_f16x2")
        out = torch.empty(B, n_classes, dtype=torch.float32, device=dev)
        scores = torch.empty(B, N, dtype=torch.float32, device=dev) if need_bwd else None
        lse = torch.empty(B, dtype=torch.float32, device=dev) if need_bwd else None
        weights = torch.empty(B, N, dtype=torch.float32, device=dev) if want_weights else None
        ls = None if logit_scale is None else _f32c(logit_scale)
        ws_bytes = _fwd_ws_bytes(lib, B, N, d, n_classes)
        st = _stream(qc)
        ws = _workspace(ws_bytes, dev, st) if ws_bytes else None
        opts = cache.call_opts(sy, n_classes, sy_call=syc) if cache is not None else _default_opts()
        with _OnDevice(dev):
            rc = lib.nw_fwd_f32(_ptr(qc), _ptr(sc), _ptr(syc), _ptr(sn2), _ptr(ssplit), _ptr(sscale), _ptr(out),
                                _ptr(scores), _ptr(lse),
                                _ptr(weights), _ptr(ws), ws_bytes, B, N, d, n_classes, kind_id,
                                _ptr(ls), int(sup_b), int(lab_b), opts, st)
        if rc:
            _lib.check(rc, "nw_fwd_f32")
        if need_bwd:
            ctx.save_for_backward(qc, sc, syc, scores, lse, out, ls if ls is not None else torch.empty(0, device=dev))
            ctx.meta = (B, N, d, n_classes, kind_id, sup_b, lab_b, ls is not None)
            # the bank of these supports, if there is one (a SplitBank's tensors are never written again)
            ctx.bank = (sn2, ssplit, sscale) if (ssplit is not None and not sup_b) else (None, None, None)
        if want_weights:
            ctx.mark_non_differentiable(weights)
            return out, weights
        return out

    @staticmethod
    def backward(ctx, gout, *unused):
        lib = _lib.load()
        qc, sc, syc, scores, lse, out, ls = ctx.saved_tensors
        B, N, d, C, kind_id, sup_b, lab_b, has_ls = ctx.meta
        dev = qc.device
        g = _f32c(gout)
        gq = torch.empty_like(qc)
        gs = torch.empty_like(sc)
        gls = torch.empty((), dtype=torch.float32, device=dev) if has_ls else None
        ws_bytes = lib.nw_bwd_workspace_bytes(B, N, d, C, kind_id, int(sup_b))
        ws = _workspace(ws_bytes, dev)
        bn2, bsplit, bscale = ctx.bank
        with torch.cuda.device(dev):
            _lib.check(lib.nw_bwd_bank_f32(_ptr(qc), _ptr(sc), _ptr(bn2), _ptr(bsplit), _ptr(bscale), _ptr(syc),
                                           _ptr(scores), _ptr(lse), _ptr(out),
                                           _ptr(g), _ptr(gq), _ptr(gs), _ptr(gls), _ptr(ws), ws_bytes,
                                           B, N, d, C, kind_id, _ptr(ls) if has_ls else None,
                                           int(sup_b), int(lab_b), _stream(qc)), "nw_bwd_bank_f32")
        return gq, gs, None, gls, None, None, None, None, None


def nw_head(q, s, sy, n_classes, kind="euclidean", logit_scale=None, return_weights=False,
            support_norm2=None, support_cache=None, validate_labels=False):
    """NWHead.forward(x, sx, sy) -> (B,C) log-probs (and the (B,N) softmax weights on request).
    support_norm2: optional cached ``row_norm2(s)`` for a shared (N,d) support.
    support_cache: optional ``SplitBank(s)`` (norms + split-fp16 rows: the fast 'full' inference path).
    validate_labels: the reference's F.one_hot (nw.py:276) REFUSES labels outside [0, n_classes); the kernels skip such
    supports silently (banks with labels check once, when they are built).  True: check here and raise F.one_hot's
    RuntimeError -- one device round trip per call, which is why it is opt-in (NWHead.validate_labels; NWNet's debug_mode
    turns it on)."""
    if validate_labels and sy.numel():
        lo, hi = (int(v) for v in torch.aminmax(sy.detach()))
        if lo < 0:
            raise RuntimeError("Class values must be non-negative.")                  # F.one_hot's own messages
        if hi >= int(n_classes):
            raise RuntimeError("Class values must be smaller than num_classes.")
    kid = _kind_id(kind)
    if kid == SCORE_KINDS["clip"] and logit_scale is None:
        raise ValueError("clip kernel needs logit_scale")
    if s.dim() == 2 and sy.dim() != 1 or s.dim() == 3 and sy.dim() != 2:
        raise ValueError("support labels must be (N,) for (N,d) supports and (B,N) for (B,N,d)")
    if support_norm2 is not None:
        if s.dim() != 2 or support_norm2.shape != (s.shape[0],):
            raise ValueError("support_norm2 must be (N,) for an (N,d) support")
        support_norm2 = _f32c(support_norm2)
    if support_cache is not None and (s.dim() != 2 or not support_cache.matches(s)):
        raise ValueError("support_cache was prepared from another support tensor (or the tensor was modified in place "
                         "since): build a new ops.SplitBank(s) -- the split rows and norms it holds are those of the "
                         "tensor it was built from")
    if support_cache is not None and support_cache.label_max is not None and support_cache.label_max >= int(n_classes):
        raise ValueError(f"support label {support_cache.label_max} is outside [0, n_classes={int(n_classes)}) "
                         "(the reference's F.one_hot, nw.py:276, raises)")
    s, sy, support_cache = _resolve_sorted_bank(s, sy, support_cache,
                                                return_weights or (torch.is_grad_enabled() and s.requires_grad))
    if support_cache is not None and support_cache.pad:
        if torch.is_grad_enabled() and s.requires_grad:    # the padded copy is not part of the caller's graph
            support_norm2, support_cache = support_cache.norm2, None
        else:
            q, s = _apply_bank_padding(q, s, support_cache)
    if q.shape[-1] % 4 and s.dim() == 2 and s.shape[0] > 25:
        # an embedding size that is not a multiple of 4 would miss every tile kernel (they move 16-byte pieces) and land
        # on the generic two-kernel path (measured at d = 130: 925 us against 19 at d = 128; 52 ms with 20000 classes):
        # zero columns change no dot product and no norm, so the operands are padded (torch ops: autograd slices the
        # gradients back) and the bank's cached norms stay valid
        pad = (-q.shape[-1]) % 4
        if support_cache is not None:
            support_norm2, support_cache = support_cache.norm2, None
        q, s = torch.nn.functional.pad(q, (0, pad)), torch.nn.functional.pad(s, (0, pad))
    needs_grad = torch.is_grad_enabled() and (q.requires_grad or s.requires_grad or
                                              (logit_scale is not None and logit_scale.requires_grad))
    d_now = q.shape[-1]
    if needs_grad:
        _lib.sync_knobs()              # (NW_BWD_SPLIT and the other diagnostic switches may be flipped between steps)
    if (needs_grad and support_cache is None and s.dim() == 2 and d_now % 32 and d_now >= 256
            and _lib.load().nw_bwd_uses_split(q.shape[0], s.shape[0], d_now + (-d_now) % 32, int(n_classes), 0)):
        # a training step at a width that is not a multiple of 32: zero columns take it to the split-row kernels of the
        # forward and the backward (d = 1000 at T: 326 -> 237 us per eager forward + backward; narrow widths lose more to the
        # two extra torch ops than the kernels gain: d = 100, 168 -> 232 us); autograd slices the gradients back
        pad = (-d_now) % 32
        q, s = torch.nn.functional.pad(q, (0, pad)), torch.nn.functional.pad(s, (0, pad))
    if not needs_grad:   # inference: skip the autograd node (its bookkeeping costs more than the kernels at small sizes)
        return _NWHeadFn.forward(_NoCtx, q, s, sy, logit_scale, int(n_classes), kid, bool(return_weights),
                                 support_norm2, support_cache)
    return _NWHeadFn.apply(q, s, sy, logit_scale, int(n_classes), kid, bool(return_weights), support_norm2,
                           support_cache)


def nw_partials(q, s, sy, n_classes, kind="euclidean", logit_scale=None, support_cache=None):
    """This rank's (m, den, num) over its shard of the bank; no grad.  ``support_cache``: the
    shard's SplitBank (split-fp16 fast path)."""
    _need_hip(q, s, sy, logit_scale)
    lib = _lib.load()
    qc, sc = _f32c(q), _f32c(s)
    syc = sy.detach().to(torch.int64).contiguous()
    B, d = qc.shape
    N = sc.shape[0]
    dev = qc.device
    packed = torch.empty(B, n_classes + 2, dtype=torch.float32, device=dev)
    sc, syc2, support_cache = _resolve_sorted_bank(sc, sy, support_cache)
    if syc2 is not sy:
        syc = syc2
    return nw_partials_into(packed, qc, sc, syc, n_classes, kind, logit_scale, cache=support_cache)


def nw_partials_into(packed, qc, sc, syc, n_classes, kind="euclidean", logit_scale=None, ws=None, sn2=None,
                     cache=None, persistent_wgs=0):
    """Write partials into ``packed`` laid out as [m (B) | den (B) | num (B*C)] (flat, contiguous):
    one buffer = one collective.  Inputs must already be fp32/int64 contiguous HIP tensors."""
    lib = _lib.load()
    qc, sc = _apply_bank_padding(qc, sc, cache)
    B, d = qc.shape
    N = sc.shape[0]
    C = int(n_classes)
    flat = packed.view(-1)
    m, den, num = flat[:B], flat[B:2 * B], flat[2 * B:2 * B + B * C]
    ws_bytes = lib.nw_fwd_workspace_bytes(B, N, d, C)
    if ws is None or ws.numel() < ws_bytes:
        ws = _workspace(ws_bytes, qc.device)
    ls = None if logit_scale is None else _f32c(logit_scale)
    ssplit = sscale = None
    if cache is not None:
        if cache.sorted_rows is not None and sc.data_ptr() != cache.sorted_rows.data_ptr():
            raise ValueError("this SplitBank holds a class-sorted copy of its support: pass cache.sorted_rows / "
                             "cache.sorted_labels (or call nw_partials, which does)")
        sn2, ssplit, sscale = cache.norm2, cache.split, cache.scale
    opts = cache.call_opts(syc, C, persistent_wgs, sy_call=syc) if cache is not None else _default_opts(persistent_wgs)
    with torch.cuda.device(qc.device):
        _lib.check(lib.nw_fwd_partial_f32(_ptr(qc), _ptr(sc), _ptr(syc), _ptr(sn2), _ptr(ssplit), _ptr(sscale),
                                          _ptr(m), _ptr(den), _ptr(num),
                                          _ptr(ws), ws.numel(), B, N, d, C, _kind_id(kind), _ptr(ls), opts,
                                          _stream(qc)), "nw_fwd_partial_f32")
    return packed


def nw_merge(packed_all, B, n_classes, out=None, class_lo=None, c_local=None):
    """packed_all: (G, L) all-gathered partial buffers, each row [m (B) | den (B) | num (B*CL)] -> (B,C)
    log-probs.  CL = n_classes, or ``c_local`` when every shard only carries the class window
    [class_lo[g], class_lo[g] + c_local).  The merge kernel reads the sections in place."""
    _need_hip(packed_all, class_lo)
    lib = _lib.load()
    assert packed_all.dim() == 2 and packed_all.is_contiguous() and packed_all.dtype == torch.float32
    G, L = packed_all.shape
    C = int(n_classes)
    CL = C if class_lo is None else int(c_local)
    assert L >= 2 * B + B * CL
    dev = packed_all.device
    if out is None:
        out = torch.empty(B, C, dtype=torch.float32, device=dev)
    base = packed_all.data_ptr()
    with torch.cuda.device(dev):
        _lib.check(lib.nw_merge_finalize_f32(base, base + 4 * B, base + 8 * B, _ptr(out), G, B, C, L, L, L,
                                             _ptr(class_lo), CL, _stream(packed_all)), "nw_merge_finalize_f32")
    return out


def scale_shift_relu(x, scale, shift, relu=True):
    """out = max(x * scale[c] + shift[c], 0) for an (n, c, h, w) fp32 activation whose (c, h, w) part is
    contiguous (a channel prefix of a wider slab is fine): eval-mode BatchNorm + ReLU in one pass."""
    _need_hip(x, scale, shift)
    n, c, h, w = x.shape
    hw = h * w
    if x.dtype != torch.float32 or (n * c * hw and (x.stride(3) != 1 or x.stride(2) != w or x.stride(1) != hw)):
        x = x.float().contiguous()
    out = torch.empty(n, c, h, w, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().nw_scale_shift_relu_f32(_ptr(x), _ptr(_f32c(scale)), _ptr(_f32c(shift)), _ptr(out), n, c, hw,
                                                       x.stride(0) if n > 1 else c * hw, int(bool(relu)), _stream(x)),
                   "nw_scale_shift_relu_f32")
    return out


def bias_act_nhwc_(x, bias, residual=None, relu=True):
    """IN PLACE x <- act(x + bias[c] (+ residual)) for an (n, c, h, w) fp32 activation in channels_last memory format
    (c % 4 == 0): the folded BatchNorm bias, a ResNet block's identity and the ReLU behind a bias-free convolution in one
    pass.  Returns x."""
    _need_hip(x, bias, residual)
    n, c, h, w = x.shape
    ok = lambda t: t.dtype == torch.float32 and t.shape == x.shape and t.is_contiguous(memory_format=torch.channels_last)
    if not ok(x) or (residual is not None and not ok(residual)) or c % 4:
        raise ValueError("bias_act_nhwc_: fp32 channels_last (n, c, h, w) tensors with c % 4 == 0")
    with _OnDevice(x.device):
        _lib.check(_lib.load().nw_bias_act_nhwc_f32(_ptr(x), _ptr(_f32c(bias)), _ptr(residual), int(bool(relu)), _ptr(x),
                                                    n * h * w, c, _stream(x)), "nw_bias_act_nhwc_f32")
    return x


def scale_shift_relu_avgpool2(x, scale, shift, relu=True):
    """avg_pool2d(max(x * scale[c] + shift[c], 0), 2) in one pass: (n, c, h, w) -> (n, c, h // 2, w // 2)."""
    _need_hip(x, scale, shift)
    x, bstride = _plane_view(x)
    n, c, h, w = x.shape
    out = torch.empty(n, c, h // 2, w // 2, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().nw_scale_shift_relu_avgpool2_f32(_ptr(x), _ptr(_f32c(scale)), _ptr(_f32c(shift)), _ptr(out), n, c, h, w,
                                                                bstride, int(bool(relu)), _stream(x)),
                   "nw_scale_shift_relu_avgpool2_f32")
    return out


def conv1x1(x, weight_t, bias=None, pre_scale=None, pre_shift=None, pre_relu=False, post_relu=False, cin=None):
    """out = post(bias + W . pre(x)) for an (n, c, h, w) fp32 activation whose (c, h, w) part is contiguous (a channel
    prefix of a wider slab is fine): pre = optional per-channel scale/shift (eval-mode BatchNorm) + ReLU, W given
    transposed as weight_t (cin rounded up to 16 with zero rows, cout), post = optional ReLU.  One matrix-core
    kernel (nw_conv1x1_f32)."""
    _need_hip(x, weight_t, bias, pre_scale, pre_shift)
    x, bstride = _plane_view(x)
    n, c, h, w = x.shape
    cin = c if cin is None else int(cin)
    cout = weight_t.shape[1]
    if c != cin or weight_t.shape[0] != (cin + 15) // 16 * 16 or not weight_t.is_contiguous():
        raise ValueError(f"conv1x1: input has {c} channels, weight_t must be ({(c + 15) // 16 * 16}, cout) contiguous")
    lib = _lib.load()
    out = torch.empty(n, cout, h, w, dtype=torch.float32, device=x.device)
    ws_bytes = lib.nw_conv1x1_workspace_bytes(n, cin, cout, h * w)
    ws = _workspace(ws_bytes, x.device) if ws_bytes else None
    with torch.cuda.device(x.device):
        _lib.check(lib.nw_conv1x1_f32(_ptr(x), bstride, _ptr(pre_scale), _ptr(pre_shift), int(bool(pre_relu)),
                                      _ptr(weight_t), _ptr(bias), int(bool(post_relu)), _ptr(out), cout * h * w,
                                      _ptr(ws), ws_bytes, n, cin, cout, h * w, _stream(x)), "nw_conv1x1_f32")
    return out


def conv3x3_weight(w):
    """(cout, cin, 3, 3) -> the operand of conv3x3: (ceil(cin / 8), 9, 8, cout), channels past cin zero."""
    cout, cin = w.shape[:2]
    pad = (-cin) % 8
    if pad:
        w = torch.cat((w, w.new_zeros(cout, pad, 3, 3)), 1)
    return w.reshape(cout, (cin + pad) // 8, 8, 9).permute(1, 3, 2, 0).contiguous()


def conv3x3(x, weight_t, cin, bias=None, residual=None, post_relu=False, out=None):
    """3x3 convolution, stride 1, padding 1, of an (n, c, h, w) fp32 activation whose (c, h, w) part is contiguous (a
    channel prefix of a slab is fine), weight_t from conv3x3_weight; bias / residual / ReLU behind it in the same
    kernel; `out`: an (n, cout, h, w) view with contiguous (cout, h, w) part to write into (a channel window of a slab)."""
    _need_hip(x, weight_t, bias, residual, out)
    x, bstride = _plane_view(x)
    n, c, h, w = x.shape
    cout = weight_t.shape[3]
    if c != cin or tuple(weight_t.shape[:3]) != ((cin + 7) // 8, 9, 8) or not weight_t.is_contiguous():
        raise ValueError("conv3x3: weight_t must be conv3x3_weight(w) of a (cout, cin, 3, 3) weight with cin = x's channels")
    if out is None:
        out = torch.empty(n, cout, h, w, dtype=torch.float32, device=x.device)
    elif out.shape != (n, cout, h, w) or out.dtype != torch.float32 or (out.numel() and (
            out.stride(3) != 1 or out.stride(2) != w or out.stride(1) != h * w)):
        raise ValueError("conv3x3: `out` must be (n, cout, h, w) fp32 with a contiguous (cout, h, w) part")
    obs = out.stride(0) if n > 1 else cout * h * w
    rbs = 0
    if residual is not None:
        residual, rbs = _plane_view(residual)
        if residual.shape != (n, cout, h, w):
            raise ValueError("conv3x3: residual must have the output's shape")
    lib = _lib.load()
    ws_bytes = lib.nw_conv3x3_workspace_bytes(n, cin, cout, h, w)   # partial tiles when K is split over workgroups (7x7 planes)
    ws = _workspace(ws_bytes, x.device) if ws_bytes else None
    with _OnDevice(x.device):
        _lib.check(lib.nw_conv3x3_f32(_ptr(x), bstride, _ptr(weight_t), _ptr(bias), _ptr(residual), rbs, int(bool(post_relu)),
                                      _ptr(out), obs, _ptr(ws), ws_bytes, n, cin, cout, h, w, _stream(x)), "nw_conv3x3_f32")
    return out


def pad_rows16(w_t):
    """(cin, cout) -> (cin rounded up to 16, cout) with zero rows: the weight operand of conv1x1."""
    cin = w_t.shape[0]
    pad = (-cin) % 16
    return w_t.contiguous() if pad == 0 else torch.cat((w_t, w_t.new_zeros(pad, w_t.shape[1]))).contiguous()


def _plane_view(x):
    """(n, c, h, w) fp32 with a contiguous (c, h, w) part (batch stride free) -> (tensor, batch stride)."""
    n, c, h, w = x.shape
    if x.dtype != torch.float32 or (x.numel() and (x.stride(3) != 1 or x.stride(2) != w or x.stride(1) != h * w)):
        x = x.float().contiguous()
    return x, (x.stride(0) if n > 1 else c * h * w)


class _BNReLUTrainFn(torch.autograd.Function):
    """relu(batch_norm(x) [+ residual]) in training mode: nw_bn_relu_train_fwd_f32 / _bwd_f32 (one kernel each)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, bn, relu, passthrough=False):
        lib = _lib.load()
        xv, bstride = _plane_view(x.detach())
        n, c, h, w = xv.shape
        rv = None
        if residual is not None:
            rv = residual.detach()
            if rv.shape != xv.shape:
                raise ValueError("residual must have the shape of x")
            rv = rv if (rv.dtype == torch.float32 and rv.is_contiguous()) else rv.float().contiguous()
        y = torch.empty(n, c, h, w, dtype=torch.float32, device=xv.device)
        mean = torch.empty(c, dtype=torch.float32, device=xv.device)
        invstd = torch.empty_like(mean)
        track = bn.track_running_stats and bn.running_mean is not None
        momentum, nbt = 0.0, None
        if track:
            if bn.momentum is None:                 # cumulative moving average: the factor needs the count on the host
                bn.num_batches_tracked += 1
                momentum = 1.0 / float(bn.num_batches_tracked)
            else:
                momentum, nbt = float(bn.momentum), bn.num_batches_tracked   # counted inside the kernel
        wc, bc = _f32c(weight), _f32c(bias)
        with torch.cuda.device(xv.device):
            _lib.check(lib.nw_bn_relu_train_fwd_f32(_ptr(xv), _ptr(rv), _ptr(wc), _ptr(bc),
                                                    _ptr(bn.running_mean) if track else None,
                                                    _ptr(bn.running_var) if track else None, _ptr(y), _ptr(mean),
                                                    _ptr(invstd), _ptr(nbt), n, c, h * w, bstride, momentum, float(bn.eps),
                                                    int(relu), _stream(xv)), "nw_bn_relu_train_fwd_f32")
        ctx.save_for_backward(xv, wc, bc, mean, invstd, *(() if rv is None else (rv,)))
        ctx.relu, ctx.bstride, ctx.passthrough = relu, bstride, passthrough
        if passthrough:                 # (y, x): x leaves again so that it has ONE consumer; its other gradient
            ctx.set_materialize_grads(False)   # arrives in backward and is added inside the kernel
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, gy, gpass=None):
        lib = _lib.load()
        xv, wc, bc, mean, invstd, *rest = ctx.saved_tensors
        rv = rest[0] if rest else None
        n, c, h, w = xv.shape
        if gy is None:                  # only the pass-through output was used
            return gpass, None, None, None, None, None, None
        gy = _f32c(gy)
        acc, acc_bs = None, 0
        if gpass is not None:
            acc, acc_bs = _plane_view(gpass)
        dx = torch.empty(n, c, h, w, dtype=torch.float32, device=xv.device)
        dr = torch.empty_like(dx) if rv is not None else None
        dg, db = torch.empty_like(mean), torch.empty_like(mean)
        with torch.cuda.device(xv.device):
            _lib.check(lib.nw_bn_relu_train_bwd_f32(_ptr(xv), _ptr(rv), _ptr(gy), _ptr(wc), _ptr(bc), _ptr(mean),
                                                    _ptr(invstd), _ptr(dx), _ptr(dr), _ptr(dg), _ptr(db), _ptr(acc), acc_bs,
                                                    n, c, h * w, ctx.bstride, int(ctx.relu), _stream(xv)),
                       "nw_bn_relu_train_bwd_f32")
        return dx, dg, db, dr, None, None, None


def bn_relu_train(x, bn, relu=True, residual=None, passthrough=False):
    """relu(bn(x)) -- or relu(bn(x) + residual), the tail of a ResNet block -- for a BatchNorm2d in training mode
    (batch statistics, running statistics updated), fp32 NCHW on the MI355X.
    passthrough=True returns (y, x'): x' is x again, to be used by x's OTHER consumer, so that the gradient
    coming back through it is added to dx inside the backward kernel."""
    _need_hip(x, bn.weight, bn.bias, residual)
    if not (bn.affine and bn.weight is not None):
        raise ValueError("bn_relu_train needs an affine BatchNorm2d")
    if x.shape[0] * x.shape[2] * x.shape[3] <= 1:      # torch.nn.functional.batch_norm's own check
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
    if bn.running_mean is not None and (bn.running_mean.dtype != torch.float32 or not bn.running_mean.is_contiguous()):
        raise ValueError("running statistics must be contiguous fp32")
    return _BNReLUTrainFn.apply(x, bn.weight, bn.bias, residual, bn, bool(relu), bool(passthrough))


class _AggregateFn(torch.autograd.Function):
    """softmax over supports -> label aggregation -> log(. + 1e-12) of a given score matrix (nw.py:285-289)."""

    @staticmethod
    def forward(ctx, scores, sy, n_classes):
        lib = _lib.load()
        sc = _f32c(scores)
        syc = sy if (sy.dtype == torch.int64 and sy.is_contiguous()) else sy.detach().to(torch.int64).contiguous()
        B, N = sc.shape
        out = torch.empty(B, n_classes, dtype=torch.float32, device=sc.device)
        lse = torch.empty(B, dtype=torch.float32, device=sc.device)
        with torch.cuda.device(sc.device):
            _lib.check(lib.nw_aggregate_f32(_ptr(sc), _ptr(syc), _ptr(out), _ptr(lse), None, B, N, n_classes,
                                            int(syc.dim() == 2), _stream(sc)), "nw_aggregate_f32")
        ctx.save_for_backward(sc, syc, lse, out)
        ctx.C = n_classes
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        sc, syc, lse, out = ctx.saved_tensors
        B, N = sc.shape
        g = _f32c(gout)
        gs = torch.empty_like(sc)
        with torch.cuda.device(sc.device):
            _lib.check(lib.nw_aggregate_bwd_f32(_ptr(sc), _ptr(syc), _ptr(lse), _ptr(out), _ptr(g), _ptr(gs), B, N, ctx.C,
                                                int(syc.dim() == 2), _stream(sc)), "nw_aggregate_bwd_f32")
        return gs, None, None


def nw_aggregate(scores, sy, n_classes):
    """(B,N) scores from ANY score function (on the device, possibly with a torch autograd history) + labels (N,) or
    (B,N) -> (B,C) log-probabilities: the softmax / label aggregation / log tail of NWHead.forward, differentiable
    with respect to the scores."""
    _need_hip(scores, sy)
    if scores.dim() != 2 or (sy.dim() == 1 and sy.shape[0] != scores.shape[1]) or (sy.dim() == 2 and sy.shape != scores.shape):
        raise ValueError("scores must be (B,N) and labels (N,) or (B,N)")
    return _AggregateFn.apply(scores, sy, int(n_classes))


def nw_head_influence(q, s, sy, n_classes, qy, kind="euclidean", logit_scale=None, support_cache=None):
    """NWHead.forward plus util/metric.py:23-50 on its own outputs in one call (no grad): returns
    (out (B,C) log-probabilities, infl (B,N)) with infl[b,j] = log((p - p*w_bj) / (p - w_bj*[sy_j == qy_b])),
    p = exp(out[b, qy_b]), w = the head's softmax weights -- which are never materialised: the tile kernel writes
    raw scores into the influence buffer and one in-place pass finishes them (nw_fwd_influence_f32).
    Shared (N,d) supports only."""
    _need_hip(q, s, sy, qy, logit_scale)
    kid = _kind_id(kind)
    if kid == SCORE_KINDS["clip"] and logit_scale is None:
        raise ValueError("clip kernel needs logit_scale")
    if s.dim() != 2 or sy.dim() != 1:
        raise ValueError("nw_head_influence takes a shared (N,d) support with (N,) labels")
    if support_cache is not None and not support_cache.matches(s):
        raise ValueError("support_cache was prepared from another support tensor")
    # influences are indexed by support position: an unsorted bank's class-sorted copy cannot serve them
    s, sy, support_cache = _resolve_sorted_bank(s, sy, support_cache, per_position_outputs=True)
    q, s = _apply_bank_padding(q, s, support_cache)
    lib = _lib.load()
    qc, sc = _f32c(q), _f32c(s)
    syc = sy.detach().to(torch.int64).contiguous()
    qyc = qy.detach().to(torch.int64).contiguous()
    B, d = qc.shape
    N, C = sc.shape[0], int(n_classes)
    if qyc.shape != (B,):
        raise ValueError("qy must be (B,)")
    dev = qc.device
    out = torch.empty(B, C, dtype=torch.float32, device=dev)
    infl = torch.empty(B, N, dtype=torch.float32, device=dev)
    sn2 = ssplit = sscale = None
    if support_cache is not None:
        sn2, ssplit, sscale = support_cache.norm2, support_cache.split, support_cache.scale
    ls = None if logit_scale is None else _f32c(logit_scale)
    ws_bytes = lib.nw_fwd_workspace_bytes(B, N, d, C)
    ws = _workspace(ws_bytes, dev)
    with torch.cuda.device(dev):
        _lib.check(lib.nw_fwd_influence_f32(_ptr(qc), _ptr(sc), _ptr(syc), _ptr(sn2), _ptr(ssplit), _ptr(sscale), _ptr(qyc),
                                            _ptr(out), None, _ptr(infl), _ptr(ws), ws.numel(), B, N, d, C, kid, _ptr(ls),
                                            _default_opts(), _stream(qc)), "nw_fwd_influence_f32")
    return out, infl


def support_influence_idx(probs, qy, w, sy):
    """Index-label form of util/metric.py:23-50: probs (B,C), qy (B,), w (B,N), sy (N,) -> (B,N)."""
    _need_hip(probs, qy, w, sy)
    lib = _lib.load()
    probs, w = _f32c(probs), _f32c(w)
    qy = qy.detach().to(torch.int64).contiguous()
    sy = sy.detach().to(torch.int64).contiguous()
    B, Cc = probs.shape
    N = w.shape[1]
    out = torch.empty(B, N, dtype=torch.float32, device=probs.device)
    with torch.cuda.device(probs.device):
        _lib.check(lib.nw_support_influence_f32(_ptr(probs), _ptr(qy), _ptr(w), _ptr(sy), _ptr(out), B, N, Cc,
                                                _stream(probs)), "nw_support_influence_f32")
    return out


# ------------------------------------------------------------------------------------------------------------------
# Convolutions on the fp16 matrix cores at fp32-grade accuracy (csrc/conv_nhwc.hip): fp32 channels_last activations,
# weights split once per update, activations split in flight with one power of two per tensor.
AMAX_SLOTS = 256    # floats of an `amax` record: per-workgroup partial maxima of its producer (their max bounds max|x|)


def absmax(x):
    """A bound on max |x| of a dense fp32 HIP tensor as an amax record -- AMAX_SLOTS partial maxima, no atomics, nothing to
    clear -- (nw_absmax_f32): what a tensor needs before it can feed conv2d_nhwc when its producer did not leave one."""
    _need_hip(x)
    xc = x if (x.dtype == torch.float32 and (x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last))) \
        else x.float().contiguous()
    out = torch.empty(AMAX_SLOTS, dtype=torch.float32, device=xc.device)
    with _OnDevice(xc.device):
        _lib.check(_lib.load().nw_absmax_f32(_ptr(xc), xc.numel(), _ptr(out), _stream(xc)), "nw_absmax_f32")
    return out


def to_nhwc_pad(x, cpad=4):
    """(n, c, h, w) fp32 HIP tensor, NCHW-contiguous or channels_last -> (n, cpad, h, w) channels_last with zero channels
    behind the c real ones, carrying `.nw_amax` (one pass: nw_to_nhwc_pad_f32).  What the stem convolution reads."""
    _need_hip(x)
    n, c, h, w = x.shape
    if x.dtype != torch.float32 or not (x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last)):
        x = x.float().contiguous()
    y = torch.empty((n, cpad, h, w), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    am = torch.empty(AMAX_SLOTS, dtype=torch.float32, device=x.device)
    with _OnDevice(x.device):
        _lib.check(_lib.load().nw_to_nhwc_pad_f32(_ptr(x), _ptr(y), _ptr(am), n, c, h * w, cpad, c * h * w, x.stride(1),
                                                  x.stride(3), _stream(x)), "nw_to_nhwc_pad_f32")
    y.nw_amax = am
    return y


class SplitConvWeight:
    """A (Cout, Cin, KH, KW) convolution weight prepared for conv2d_nhwc: its channels_last bytes are a (Cout,
    KH*KW*Cin) matrix whose rows go through nw_split_rows_f16x2 (one power of two per output channel).  Rebuild after
    every weight update."""

    def __init__(self, w):
        _need_hip(w)
        cout, cin, kh, kw = w.shape
        if cin == 3 and kw <= 8:        # RGB stems run on a 4-channel input (to_nhwc_pad): a pixel is one aligned float4
            w = torch.nn.functional.pad(w.detach(), (0, 0, 0, 0, 0, 1))
            cin = 4
        rows = w.detach().float().permute(0, 2, 3, 1).contiguous()          # (cout, kh, kw, cin)
        if cin % 32 == 0:
            rows = rows.view(cout, kh * kw * cin)
        elif kw * cin <= 32:
            # few input channels (the 7x7 stem over RGB): one kernel ROW is one 32-wide k chunk, zero-padded
            rows = torch.nn.functional.pad(rows.view(cout, kh, kw * cin), (0, 32 - kw * cin)).reshape(cout, kh * 32)
        else:
            raise ValueError("conv2d_nhwc needs Cin % 32 == 0 (or KW * Cin <= 32)")
        self.shape = (cout, cin, kh, kw)
        self.split = torch.empty_like(rows)
        self.scale = torch.empty(cout, dtype=torch.float32, device=rows.device)
        norm2 = torch.empty(cout, dtype=torch.float32, device=rows.device)
        with _OnDevice(rows.device):
            _lib.check(_lib.load().nw_split_rows_f16x2(_ptr(rows), _ptr(self.split), _ptr(self.scale), _ptr(norm2),
                                                       cout, rows.shape[1], _stream(rows)), "nw_split_rows_f16x2")


def conv2d_nhwc_supported(x_shape, w_shape, stride, pad):
    n, cin, h, w = x_shape
    cout, cin2, kh, kw = w_shape
    return cin == cin2 and bool(_lib.load().nw_conv2d_nhwc_supported(n, h, w, cin, cout, kh, kw, stride, pad))


def conv2d_nhwc(x, weight, bias=None, residual=None, relu=False, stride=1, pad=0, amax=None, want_amax=True, room=0):
    """y = post(conv2d(x, W, stride, pad) + bias [+ residual]) for a channels_last fp32 (n, Cin, H, W) HIP tensor and a
    SplitConvWeight; returns a channels_last (n, Cout, Ho, Wo) tensor.  `amax`: the amax record of x (AMAX_SLOTS floats
    whose maximum bounds max|x|; default: x.nw_amax when x came out of this function, else one absmax pass); the result
    carries its own in `.nw_amax` when want_amax.  room: channels to leave behind every pixel's Cout (_with_room)."""
    _need_hip(x, bias, residual)
    lib = _lib.load()
    n, cin, h, w = x.shape
    cout, cin2, kh, kw = weight.shape
    if cin == 3 and cin2 == 4:          # an RGB stem: SplitConvWeight padded its weight, the input follows (one pass, with amax)
        x = to_nhwc_pad(x, 4)
        cin = 4
    if cin != cin2:
        raise ValueError(f"conv2d_nhwc: input has {cin} channels, the weight {cin2}")
    if x.dtype != torch.float32 or not x.is_contiguous(memory_format=torch.channels_last):
        x = x.float().contiguous(memory_format=torch.channels_last)
    if amax is None:
        amax = getattr(x, "nw_amax", None)
        if amax is None:
            amax = absmax(x)
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    y, ldy = _with_room(n, cout, ho, wo, room, x.device)
    if residual is not None and (residual.shape != y.shape or residual.dtype != torch.float32
                                 or not residual.is_contiguous(memory_format=torch.channels_last)):
        residual = residual.float().expand_as(y).contiguous(memory_format=torch.channels_last)
    am_out = torch.empty(AMAX_SLOTS, dtype=torch.float32, device=x.device) if want_amax else None
    with _OnDevice(x.device):
        _lib.check(lib.nw_conv2d_nhwc_f16x2(_ptr(x), _ptr(amax), _ptr(weight.split), _ptr(weight.scale),
                                            None if bias is None else _ptr(_f32c(bias)), _ptr(residual), int(bool(relu)),
                                            _ptr(y), _ptr(am_out), n, h, w, cin, cout, kh, kw, int(stride), int(pad), 0,
                                            ldy if room > 0 else 0, None, _stream(x)), "nw_conv2d_nhwc_f16x2")
    if want_amax:
        y.nw_amax = am_out
    return y


STEM_POOL_FUSED = os.environ.get("NW_STEM_POOL_FUSED", "1") != "0"


@torch.no_grad()
def stem_conv_relu_maxpool_nhwc(x, weight, bias, room=0):
    """maxpool3x3/2/1(relu(conv7x7/2/3(x) + bias)) of an ImageNet stem at inference (model/resnet.py:147, :200-203,
    model/densenet.py:114-120; BatchNorm folded into `weight`, a SplitConvWeight of shape (64, 4, 7, 7), and `bias`) as ONE kernel
    (nw_stem7x7s2_relu_maxpool_f16x2) when it serves the shape, else the convolution and the pool; x: (n, 3, H, W) fp32 on the
    device.  The result is channels_last with `.nw_amax`; room: see conv2d_nhwc."""
    _need_hip(x, bias)
    lib = _lib.load()
    n, cin, h, w = x.shape
    cout = weight.shape[0]
    if not (STEM_POOL_FUSED and cin == 3 and tuple(weight.shape[1:]) == (4, 7, 7) and bias is not None
            and lib.nw_stem7x7s2_relu_maxpool_supported(n, h, w, cout)):
        return maxpool3s2_nhwc(conv2d_nhwc(x, weight, bias, None, True, 2, 3), room)
    x4 = to_nhwc_pad(x, 4)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    hp, wp = (ho - 1) // 2 + 1, (wo - 1) // 2 + 1
    y, ldy = _with_room(n, cout, hp, wp, room, x.device)
    am = torch.empty(AMAX_SLOTS, dtype=torch.float32, device=x.device)
    with _OnDevice(x.device):
        _lib.check(lib.nw_stem7x7s2_relu_maxpool_f16x2(_ptr(x4), _ptr(x4.nw_amax), _ptr(weight.split), _ptr(weight.scale), _ptr(_f32c(bias)),
                                                       _ptr(y), _ptr(am), n, h, w, ldy if room > 0 else 0, _stream(x4)),
                   "nw_stem7x7s2_relu_maxpool_f16x2")
    y.nw_amax = am
    return y


# ------------------------------------------------------------------------------------------------------------------
# Channels-last training path: BatchNorm + ReLU (csrc/bn_nhwc.hip) and the convolution's autograd node.
def _nhwc_rows(x):
    """(n, c, h, w) fp32 HIP tensor whose memory is (n h w) rows of >= c floats (channels_last, or a channel prefix /
    window of a channels_last tensor) -> (tensor, row stride); anything else is made channels_last."""
    n, c, h, w = x.shape
    if x.dtype == torch.float32 and x.numel():
        ld = x.stride(3) if w > 1 else (x.stride(2) if h > 1 else (x.stride(0) if n > 1 else c))
        if (x.stride(1) == 1 and ld >= c and ld % 4 == 0 and (w == 1 or x.stride(3) == ld) and (h == 1 or x.stride(2) == w * ld)
                and (n == 1 or x.stride(0) == h * w * ld) and x.data_ptr() % 16 == 0):
            return x, ld
    x = x.float().contiguous(memory_format=torch.channels_last)
    return x, c


class _BNReLUNhwcFn(torch.autograd.Function):
    """relu(batch_norm(x)) in training mode over channels-last activations (nw_bn_relu_nhwc_train_fwd_f32 / _bwd_f32)."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn, relu, passthrough=False):
        lib = _lib.load()
        xv, ldx = _nhwc_rows(x.detach())
        n, c, h, w = xv.shape
        rows = n * h * w
        dev = xv.device
        y = torch.empty((n, c, h, w), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        mean = torch.empty(c, dtype=torch.float32, device=dev)
        invstd = torch.empty_like(mean)
        amax = torch.empty(AMAX_SLOTS, dtype=torch.float32, device=dev)
        track = bn.track_running_stats and bn.running_mean is not None
        momentum, nbt = 0.0, None
        if track:
            if bn.momentum is None:
                bn.num_batches_tracked += 1
                momentum = 1.0 / float(bn.num_batches_tracked)
            else:
                momentum, nbt = float(bn.momentum), bn.num_batches_tracked
        wc, bc = _f32c(weight), _f32c(bias)
        ws_bytes = lib.nw_bn_nhwc_workspace_bytes(rows, c)
        ws = _workspace(ws_bytes, dev)
        with _OnDevice(dev):
            _lib.check(lib.nw_bn_relu_nhwc_train_fwd_f32(_ptr(xv), ldx, _ptr(wc), _ptr(bc),
                                                         _ptr(bn.running_mean) if track else None,
                                                         _ptr(bn.running_var) if track else None, _ptr(y), _ptr(mean),
                                                         _ptr(invstd), _ptr(nbt), _ptr(amax), _ptr(ws), ws_bytes, rows, c,
                                                         momentum, float(bn.eps), int(relu), _stream(xv)),
                       "nw_bn_relu_nhwc_train_fwd_f32")
        ctx.save_for_backward(xv, wc, bc, mean, invstd)
        ctx.relu, ctx.ldx, ctx.passthrough = relu, ldx, passthrough
        y.nw_amax = amax
        if passthrough:
            ctx.set_materialize_grads(False)
            return y, x.view_as(x)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, gpass=None):
        lib = _lib.load()
        xv, wc, bc, mean, invstd = ctx.saved_tensors
        if gy is None:
            return gpass, None, None, None, None, None
        n, c, h, w = xv.shape
        rows = n * h * w
        dev = xv.device
        if gy.dtype != torch.float32 or not gy.is_contiguous(memory_format=torch.channels_last):
            gy = gy.float().contiguous(memory_format=torch.channels_last)
        acc, ldacc = None, 0
        if gpass is not None:
            acc, ldacc = _nhwc_rows(gpass)
        dx = torch.empty((n, c, h, w), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        dg, db = torch.empty_like(mean), torch.empty_like(mean)
        amax = torch.empty(AMAX_SLOTS, dtype=torch.float32, device=dev)
        ws_bytes = lib.nw_bn_nhwc_workspace_bytes(rows, c)
        ws = _workspace(ws_bytes, dev)
        with _OnDevice(dev):
            _lib.check(lib.nw_bn_relu_nhwc_train_bwd_f32(_ptr(xv), ctx.ldx, _ptr(gy), _ptr(wc), _ptr(bc), _ptr(mean), _ptr(invstd),
                                                         _ptr(dx), _ptr(dg), _ptr(db), _ptr(acc), ldacc, 0, _ptr(amax), _ptr(ws),
                                                         ws_bytes, rows, c, int(ctx.relu), _stream(xv)),
                       "nw_bn_relu_nhwc_train_bwd_f32")
        dx.nw_amax = amax
        dx.nw_fresh = True        # nobody else holds this buffer: a dense block's backward may use it as its gradient slab
        return dx, dg, db, None, None, None


BN_NHWC_MAX_C = 2560      # channels nw_bn_relu_nhwc_train_* serve (bn_nhwc.hip: BN_MAX_C)


def bn_relu_train_nhwc(x, bn, relu=True, passthrough=False):
    """relu(bn(x)) for a BatchNorm2d in training mode over a channels-last fp32 activation on the MI355X; the result is
    channels_last and carries `.nw_amax` for the convolution that follows.  passthrough: see bn_relu_train."""
    _need_hip(x, bn.weight, bn.bias)
    if not (bn.affine and bn.weight is not None):
        raise ValueError("bn_relu_train_nhwc needs an affine BatchNorm2d")
    if x.shape[0] * x.shape[2] * x.shape[3] <= 1:
        raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
    if x.shape[1] % 4:
        raise ValueError("bn_relu_train_nhwc needs a channel count that is a multiple of 4")
    if x.shape[1] > BN_NHWC_MAX_C:           # wider than the kernels' per-channel tables: torch's BatchNorm on the same layout
        # (torch's own order: count first; momentum None = cumulative average with factor 1 / count, nn/modules/batchnorm.py)
        if bn.track_running_stats and bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        factor = bn.momentum
        if factor is None:
            factor = 1.0 / float(bn.num_batches_tracked) if (bn.track_running_stats and bn.num_batches_tracked is not None) else 0.0
        y = torch.nn.functional.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, True, factor, bn.eps)
        y = torch.relu(y) if relu else y
        return (y, x) if passthrough else y
    return _BNReLUNhwcFn.apply(x, bn.weight, bn.bias, bn, bool(relu), bool(passthrough))


class _AddReluNhwcFn(torch.autograd.Function):
    """relu(a + b) over two channels-last tensors (the end of a residual block, model/resnet.py:60-66) with the amax record of the
    result; backward: ONE masked gradient for both summands (nw_add_relu_f32 / nw_relu_bwd_f32)."""

    @staticmethod
    def forward(ctx, a, b):
        lib = _lib.load()
        av, bv = a.detach(), b.detach()
        if not (av.is_contiguous(memory_format=torch.channels_last) and bv.is_contiguous(memory_format=torch.channels_last)):
            av, bv = av.contiguous(memory_format=torch.channels_last), bv.contiguous(memory_format=torch.channels_last)
        out = torch.empty_like(av, memory_format=torch.channels_last)
        amax = torch.empty(AMAX_SLOTS, dtype=torch.float32, device=av.device)
        with _OnDevice(av.device):
            _lib.check(lib.nw_add_relu_f32(_ptr(av), _ptr(bv), _ptr(out), _ptr(amax), av.numel(), _stream(av)), "nw_add_relu_f32")
        ctx.save_for_backward(out)
        out.nw_amax = amax
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        lib = _lib.load()
        out, = ctx.saved_tensors
        if g.dtype != torch.float32 or not g.is_contiguous(memory_format=torch.channels_last):
            g = g.float().contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(out, memory_format=torch.channels_last)
        amax = torch.empty(AMAX_SLOTS, dtype=torch.float32, device=out.device)
        with _OnDevice(out.device):
            _lib.check(lib.nw_relu_bwd_f32(_ptr(out), _ptr(g), _ptr(dx), _ptr(amax), out.numel(), _stream(out)), "nw_relu_bwd_f32")
        dx.nw_amax = amax
        return dx, dx


def add_relu_nhwc(a, b):
    """relu(a + b) for two channels-last fp32 activations of one shape on the MI355X; the result carries `.nw_amax`."""
    _need_hip(a, b)
    if a.shape != b.shape or a.dtype != torch.float32 or b.dtype != torch.float32 or a.numel() % 4:
        out = torch.relu(a + b)
        out.nw_amax = absmax(out.detach())
        return out
    return _AddReluNhwcFn.apply(a, b)


def _with_room(n, c, h, w, room, dev):
    """A channels-last (n, c, h, w) result tensor, alone or -- room > 0 -- as the first c channels of a fresh (n, c + room, h, w)
    allocation it carries as `.nw_slab`: a dense block that follows adopts that allocation as its slab instead of copying its
    input into one (_DenseBlockNhwcFn).  -> (tensor, row stride)"""
    if room <= 0:
        return torch.empty((n, c, h, w), dtype=torch.float32, device=dev, memory_format=torch.channels_last), c
    full = torch.empty((n, c + room, h, w), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
    y = full[:, :c]
    y.nw_slab = full
    return y, c + room


class _AvgPool2NhwcFn(torch.autograd.Function):
    """F.avg_pool2d(x, 2, 2) over a channels-last fp32 activation (nw_avgpool2x2_nhwc_f32 / _bwd_f32)."""

    @staticmethod
    def forward(ctx, x, room):
        lib = _lib.load()
        xv, ldx = _nhwc_rows(x.detach())
        n, c, h, w = xv.shape
        y, ldy = _with_room(n, c, h // 2, w // 2, room, xv.device)
        with _OnDevice(xv.device):
            _lib.check(lib.nw_avgpool2x2_nhwc_f32(_ptr(xv), ldx, _ptr(y), ldy, n, h, w, c, _stream(xv)), "nw_avgpool2x2_nhwc_f32")
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        lib = _lib.load()
        n, c, h, w = ctx.shape
        gv, ldg = _nhwc_rows(gy)
        gx = torch.empty((n, c, h, w), dtype=torch.float32, device=gv.device, memory_format=torch.channels_last)
        with _OnDevice(gv.device):
            _lib.check(lib.nw_avgpool2x2_nhwc_bwd_f32(_ptr(gv), ldg, _ptr(gx), c, n, h, w, c, _stream(gv)),
                       "nw_avgpool2x2_nhwc_bwd_f32")
        return gx, None


class _MaxPool3s2NhwcFn(torch.autograd.Function):
    """F.max_pool2d(x, 3, 2, 1) over a channels-last fp32 activation (nw_maxpool3x3s2_nhwc_f32 / _bwd_f32): the winning tap of
    each window is kept as one byte for the backward."""

    @staticmethod
    def forward(ctx, x, room):
        lib = _lib.load()
        xv, ldx = _nhwc_rows(x.detach())
        n, c, h, w = xv.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y, ldy = _with_room(n, c, ho, wo, room, xv.device)
        tap = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=xv.device)
        with _OnDevice(xv.device):
            _lib.check(lib.nw_maxpool3x3s2_nhwc_f32(_ptr(xv), ldx, _ptr(y), ldy, _ptr(tap), n, h, w, c, _stream(xv)),
                       "nw_maxpool3x3s2_nhwc_f32")
        ctx.save_for_backward(tap)
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        lib = _lib.load()
        tap, = ctx.saved_tensors
        n, c, h, w = ctx.shape
        gv, ldg = _nhwc_rows(gy)
        gx = torch.empty((n, c, h, w), dtype=torch.float32, device=gv.device, memory_format=torch.channels_last)
        with _OnDevice(gv.device):
            _lib.check(lib.nw_maxpool3x3s2_nhwc_bwd_f32(_ptr(gv), ldg, _ptr(tap), _ptr(gx), c, n, h, w, c, _stream(gv)),
                       "nw_maxpool3x3s2_nhwc_bwd_f32")
        return gx, None


def avgpool2_nhwc(x, room=0):
    """nn.AvgPool2d(2, 2) of a DenseNet transition (reference model/densenet.py:83-91) over a channels-last fp32 activation on
    the MI355X; the result is channels_last.  room: channels to leave behind every pixel's c (see _with_room)."""
    _need_hip(x)
    if x.dim() != 4 or x.shape[1] % 4 or x.shape[2] < 2 or x.shape[3] < 2:
        raise ValueError(f"avgpool2_nhwc needs (n, c % 4 == 0, h >= 2, w >= 2), got {tuple(x.shape)}")
    y = _AvgPool2NhwcFn.apply(x, int(room))
    if hasattr(x, "nw_amax"):           # an average is no larger than the largest value averaged: x's bound holds for y
        y.nw_amax = x.nw_amax
    return y


def maxpool3s2_nhwc(x, room=0):
    """nn.MaxPool2d(3, 2, 1) of the stems (reference model/densenet.py:114, model/resnet.py:147) over a channels-last fp32
    activation on the MI355X; the result is channels_last and keeps x's `.nw_amax` bound."""
    _need_hip(x)
    if x.dim() != 4 or x.shape[1] % 4:
        raise ValueError(f"maxpool3s2_nhwc needs (n, c % 4 == 0, h, w), got {tuple(x.shape)}")
    if not (torch.is_grad_enabled() and x.requires_grad):       # inference: no tap record
        xv, ldx = _nhwc_rows(x.detach())
        n, c, h, w = xv.shape
        y, ldy = _with_room(n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1, int(room), xv.device)
        with _OnDevice(xv.device):
            _lib.check(_lib.load().nw_maxpool3x3s2_nhwc_f32(_ptr(xv), ldx, _ptr(y), ldy, None, n, h, w, c, _stream(xv)),
                       "nw_maxpool3x3s2_nhwc_f32")
    else:
        y = _MaxPool3s2NhwcFn.apply(x, int(room))
    if hasattr(x, "nw_amax"):
        y.nw_amax = x.nw_amax
    return y


class _WeightOperand:
    """A split-row convolution operand inside a ConvWeightBank (the interface conv2d_nhwc reads: split, scale, shape)."""
    __slots__ = ("split", "scale", "shape")

    def __init__(self, split, scale, shape):
        self.split, self.scale, self.shape = split, scale, shape


class ConvWeightBank:
    """The split-row operands of ALL convolution weights of a network -- forward and, for stride-1 convolutions whose input
    needs a gradient, data-gradient (flipped taps, transposed channels) -- kept in two flat buffers and rebuilt by ONE
    launch (nw_split_conv_weights_f16x2) whenever a weight has changed since the last build (an optimizer step)."""

    def __init__(self, convs):
        """convs: list of (weight parameter (Cout, Cin, KH, KW), needs_dgrad)."""
        self.weights = [w for w, _ in convs]
        self._want = [bool(d) for _, d in convs]
        self._sig = None
        self._build_tables()

    def _build_tables(self):
        dev = self.weights[0].device
        jobs, views = [], []
        off = soff = rows_total = 0
        for w, want_dgrad in zip(self.weights, self._want):
            cout, cin, kh, kw = w.shape
            t = kh * kw
            if not (w.is_contiguous() and w.dtype == torch.float32):
                raise ValueError("ConvWeightBank needs contiguous fp32 weights")
            entry = [None, None]
            if cin % 32 == 0:
                rows, cols, mode, shape = cout, t * cin, 0, (cout, cin, kh, kw)
            elif cin <= 4 and kw * 4 <= 32:
                rows, cols, mode, shape = cout, kh * 32, 2, (cout, 4, kh, kw)
            else:
                rows = None
            if rows is not None:
                jobs.append([w.data_ptr(), off, soff, rows_total, rows, cols, cin, cout, t, kw | (mode << 32)])
                entry[0] = (off, soff, rows, cols, shape)
                off, soff, rows_total = off + rows * cols, soff + rows, rows_total + rows
            if want_dgrad and cin % 32 == 0 and cout % 32 == 0:
                rows, cols = cin, t * cout
                jobs.append([w.data_ptr(), off, soff, rows_total, rows, cols, cin, cout, t, kw | (1 << 32)])
                entry[1] = (off, soff, rows, cols, (cin, cout, kh, kw))
                off, soff, rows_total = off + rows * cols, soff + rows, rows_total + rows
            views.append(entry)
        self.split = torch.empty(max(off, 4), dtype=torch.float32, device=dev)
        self.scale = torch.empty(max(soff, 4), dtype=torch.float32, device=dev)
        self.jobs = torch.tensor(jobs, dtype=torch.int64).to(dev)
        self.total_rows = rows_total
        self._ops = {}
        for w, entry in zip(self.weights, views):
            made = []
            for e in entry:
                if e is None:
                    made.append(None)
                else:
                    o_, so_, rows, cols, shape = e
                    made.append(_WeightOperand(self.split[o_:o_ + rows * cols].view(rows, cols), self.scale[so_:so_ + rows], shape))
            self._ops[id(w)] = tuple(made)
        self._ptrs = tuple(w.data_ptr() for w in self.weights)

    def refresh(self, force=False):
        """Rebuild the operands if any weight changed (in place: its version; replaced: its address).  force: rebuild
        regardless -- what a training forward asks for, because torch's FUSED optimizers (SGD / Adam(fused=True)) update
        parameters without advancing their version counters, so an unchanged version proves nothing there."""
        ptrs = tuple(w.data_ptr() for w in self.weights)
        if ptrs != self._ptrs:
            self._build_tables()
            self._sig = None
        sig = tuple(_ver(w) for w in self.weights)
        if sig == self._sig and not force:
            return
        dev = self.split.device
        with _OnDevice(dev):
            _lib.check(_lib.load().nw_split_conv_weights_f16x2(_ptr(self.jobs), self.jobs.shape[0], self.total_rows,
                                                               _ptr(self.split), _ptr(self.scale), _stream(self.split)),
                       "nw_split_conv_weights_f16x2")
        self._sig = sig

    def has(self, w):
        return id(w) in self._ops

    def operands(self, w):
        """(forward operand, data-gradient operand or None) of weight parameter w."""
        return self._ops[id(w)]


_CONV_STATS = {"absmax_fallbacks": 0}


def _amax_of(t):
    """The amax record a tensor carries (conv2d_nhwc / bn_relu_train_nhwc leave one on their results), else one pass."""
    a = getattr(t, "nw_amax", None)
    if a is None:
        _CONV_STATS["absmax_fallbacks"] += 1
        a = absmax(t)
    return a


# Measured and left OFF: with the weight gradient of every convolution on a side stream beside its data gradient a
# DenseNet-121 step takes 24.9-26.5 ms against 20.8-22.0 (same box, alternated): the two cross-stream dependencies per
# node (event + barrier packet each way) cost more than the overlap of two kernels that both want every CU returns.
WGRAD_SIDE_STREAM = os.environ.get("NW_WGRAD_STREAM", "0") == "1"
# Data and weight gradients of STRIDED many-channel convolutions (the ResNets' 3x3 / 2 and 1x1 / 2) on the own kernels: the data
# gradient as the stride-1 one over gy with zeros between its pixels, the weight gradient as one 1x1 problem per tap
# (round 4).  0: torch / MIOpen.
OWN_STRIDED_GRADS = os.environ.get("NW_OWN_STRIDED_GRADS", "1") != "0"
_SIDE_STREAMS = {}


def _side_stream(dev):
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _SIDE_STREAMS.get(idx)
    if s is None:
        s = _SIDE_STREAMS[idx] = torch.cuda.Stream(device=dev)
    return s


class _ConvNhwcFn(torch.autograd.Function):
    """conv2d over channels-last activations on the fp16 matrix cores: forward and data gradient through
    nw_conv2d_nhwc_f16x2 (the data gradient of a stride-1 convolution is the convolution with the flipped, transposed
    weight and padding k - 1 - p), weight gradient through nw_conv2d_nhwc_wgrad_f16x2 when it serves the shape."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad, amax, operands, room=0):
        xv = x.detach()
        if xv.dtype != torch.float32 or not xv.is_contiguous(memory_format=torch.channels_last):
            xv = xv.float().contiguous(memory_format=torch.channels_last)
        fw = operands[0] if operands is not None and operands[0] is not None else SplitConvWeight(weight)
        if amax is None and not (xv.shape[1] == 3 and fw.shape[1] == 4):   # (an RGB stem gets its record from the padding pass)
            amax = _amax_of(x)
        y = conv2d_nhwc(xv, fw, None, None, False, stride, pad, amax=amax, room=room)
        ctx.save_for_backward(xv, weight, amax if amax is not None else torch.empty(0, device=xv.device))
        ctx.stride, ctx.pad = stride, pad
        ctx.dgrad_operand = operands[1] if operands is not None else None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        xv, weight, amax_x = ctx.saved_tensors
        stride, pad = ctx.stride, ctx.pad
        cout, cin, kh, kw = weight.shape
        g = gy
        gam = getattr(gy, "nw_amax", None)
        if g.dtype != torch.float32 or not g.is_contiguous(memory_format=torch.channels_last):
            g = g.float().contiguous(memory_format=torch.channels_last)
        if gam is None:
            _CONV_STATS["absmax_fallbacks"] += 1
            gam = absmax(g)
        dx = dw = None
        # The weight gradient is off the backward's critical path (the data gradient feeds the next node): it goes to a
        # side stream and runs beside the data gradient; the main stream waits for it before the node returns, so the
        # tensors autograd sees afterwards are complete on the stream it uses.
        side = None
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[0] and WGRAD_SIDE_STREAM:
            main = torch.cuda.current_stream(g.device)
            side = _side_stream(g.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                dw = conv2d_nhwc_wgrad(xv, g, weight.shape, stride, pad, amax_x if amax_x.numel() else None, gam)
            dw.record_stream(main)
        if ctx.needs_input_grad[0]:
            if stride == 1 and kh == kw and kh - 1 - pad >= 0 and cout % 32 == 0 and cin % 32 == 0:
                dg = ctx.dgrad_operand
                if dg is None:
                    dg = SplitConvWeight(weight.detach().flip(2, 3).transpose(0, 1))     # (cin, cout, kh, kw)
                dx = conv2d_nhwc(g, dg, None, None, False, 1, kh - 1 - pad, amax=gam)
            elif (OWN_STRIDED_GRADS and stride > 1 and kh == kw and kh - 1 - pad >= 0 and cout % 32 == 0 and cin % 32 == 0
                  and xv.shape[2] + 2 * pad - kh + 1 >= (g.shape[2] - 1) * stride + 1):
                # strided: the stride-1 data gradient of gy with zeros between its pixels (model/resnet.py's 3x3 / 2 and 1x1 / 2
                # convolutions; three quarters of the products are with zeros -- six small layers of a ResNet)
                dg = ctx.dgrad_operand
                if dg is None:
                    dg = SplitConvWeight(weight.detach().flip(2, 3).transpose(0, 1))
                hd, wd = xv.shape[2] + 2 * pad - kh + 1, xv.shape[3] + 2 * pad - kw + 1
                gd = torch.zeros((g.shape[0], cout, hd, wd), dtype=torch.float32, device=g.device).contiguous(memory_format=torch.channels_last)
                gd[:, :, 0:(g.shape[2] - 1) * stride + 1:stride, 0:(g.shape[3] - 1) * stride + 1:stride] = g
                dx = conv2d_nhwc(gd, dg, None, None, False, 1, kh - 1 - pad, amax=gam)
            else:
                dx = torch.ops.aten.convolution_backward(g, xv, weight, None, [stride, stride], [pad, pad], [1, 1], False,
                                                         [0, 0], 1, [True, False, False])[0]
        if side is not None:
            main.wait_stream(side)
        elif ctx.needs_input_grad[1]:
            dw = conv2d_nhwc_wgrad(xv, g, weight.shape, stride, pad, amax_x if amax_x.numel() else None, gam)
        return dx, dw, None, None, None, None, None


def conv2d_nhwc_wgrad(x, gy, wshape, stride, pad, amax_x=None, amax_g=None):
    """Weight gradient of conv2d for channels-last x (n, Cin, H, W) and gy (n, Cout, Ho, Wo) -> (Cout, Cin, KH, KW)
    (channels_last strides).  Stride-1 'same' 1x1 / 3x3 shapes run in nw_conv2d_nhwc_wgrad_f16x2; the rest (the strided
    stem) goes to MIOpen through torch."""
    lib = _lib.load()
    cout, cin, kh, kw = wshape
    n, _, h, w = x.shape
    if lib.nw_conv2d_nhwc_wgrad_supported(n, h, w, cin, cout, kh, kw, stride, pad):
        if amax_x is None:
            amax_x = _amax_of(x)
        if amax_g is None:
            amax_g = _amax_of(gy)
        dw = torch.empty((cout, kh, kw, cin), dtype=torch.float32, device=x.device)
        ws_bytes = lib.nw_conv2d_nhwc_wgrad_workspace_bytes(n, h, w, cin, cout, kh, kw, stride, pad)
        ws = _workspace(ws_bytes, x.device)
        with _OnDevice(x.device):
            _lib.check(lib.nw_conv2d_nhwc_wgrad_f16x2(_ptr(x), _ptr(amax_x), _ptr(gy), _ptr(amax_g), _ptr(dw), _ptr(ws), ws_bytes,
                                                      n, h, w, cin, cout, kh, kw, stride, pad, 0, 0, _stream(x)),
                       "nw_conv2d_nhwc_wgrad_f16x2")
        if kh == 1 and kw == 1:
            return dw.view(cout, cin, 1, 1)      # the same bytes in torch's own strides: AccumulateGrad takes it without a copy
        return dw.permute(0, 3, 1, 2)
    if cin == 3 and 4 * kw <= 32 and cout % 8 == 0 and x.is_cuda:
        # The few-channel stems (7x7 / 2 over RGB, model/densenet.py:114-116, model/resnet.py:147; round 4, VERDICT r03 item 3c:
        # MIOpen's igemm_wrw kernel was the last vendor kernel in K4's trace, with a 5 s solver search on a process's first
        # step): ONE 1x1 weight gradient between gy and, per kernel row, the 32-float run that row reads from the 4-channel
        # padded input (nw_wgrad_job.rowrun_stride: 32 KH "channels").
        x4 = to_nhwc_pad(x, 4)
        ho, wo = gy.shape[2], gy.shape[3]
        if amax_g is None:
            amax_g = _amax_of(gy)
        gyc = gy if gy.is_contiguous(memory_format=torch.channels_last) else gy.contiguous(memory_format=torch.channels_last)
        rows = torch.empty((cout, kh * 32), dtype=torch.float32, device=x.device)
        jobs = (_lib.WgradJob * 1)(_lib.WgradJob(_ptr(x4), _ptr(x4.nw_amax), _ptr(gyc), _ptr(amax_g), _ptr(rows), n, ho, wo, 32 * kh, cout,
                                                 1, 1, 1, 0, 0, 0, 0, None, int(stride), h, w, -pad, -pad))
        wsb = lib.nw_conv2d_nhwc_wgrad_batch_workspace_bytes(jobs, 1)
        ws = _workspace(wsb, x.device)
        with _OnDevice(x.device):
            _lib.check(lib.nw_conv2d_nhwc_wgrad_batch_f16x2(jobs, 1, _ptr(ws), wsb, _stream(x)), "nw_conv2d_nhwc_wgrad_batch_f16x2")
        # rows[co, 32 ky + 4 kx + ci] -> (cout, cin, kh, kw)
        return rows.view(cout, kh, 8, 4)[:, :, :kw, :3].permute(0, 3, 1, 2)
    if OWN_STRIDED_GRADS and stride > 1 and cin % 8 == 0 and cout % 8 == 0 and x.is_cuda and \
            lib.nw_conv2d_nhwc_wgrad_supported(n, gy.shape[2], gy.shape[3], cin, cout, 1, 1, 1, 0):
        # strided convolutions with many channels (model/resnet.py:31-66: 3x3 / 2, 1x1 / 2): one 1x1 problem per tap between gy
        # and the input pixels that tap reads (a strided slice of the padded input, copied densely), batched into one launch
        ho, wo = gy.shape[2], gy.shape[3]
        if amax_x is None:
            amax_x = _amax_of(x)
        if amax_g is None:
            amax_g = _amax_of(gy)
        gyc = gy if gy.is_contiguous(memory_format=torch.channels_last) else gy.contiguous(memory_format=torch.channels_last)
        xp = torch.nn.functional.pad(x, (pad, pad, pad, pad)) if pad else x
        taps = [xp[:, :, ky:ky + stride * (ho - 1) + 1:stride, kx:kx + stride * (wo - 1) + 1:stride].contiguous(memory_format=torch.channels_last)
                for ky in range(kh) for kx in range(kw)]
        dwt = torch.empty((kh * kw, cout, cin), dtype=torch.float32, device=x.device)
        jl = [_lib.WgradJob(_ptr(t), _ptr(amax_x), _ptr(gyc), _ptr(amax_g), dwt.data_ptr() + 4 * k * cout * cin, n, ho, wo, cin, cout,
                            1, 1, 1, 0, 0, 0, 0, None) for k, t in enumerate(taps)]
        jobs = (_lib.WgradJob * len(jl))(*jl)
        wsb = lib.nw_conv2d_nhwc_wgrad_batch_workspace_bytes(jobs, len(jl))
        ws = _workspace(wsb, x.device)
        with _OnDevice(x.device):
            _lib.check(lib.nw_conv2d_nhwc_wgrad_batch_f16x2(jobs, len(jl), _ptr(ws), wsb, _stream(x)), "nw_conv2d_nhwc_wgrad_batch_f16x2")
        del taps
        return dwt.permute(1, 2, 0).reshape(cout, cin, kh, kw) if kh * kw > 1 else dwt.view(cout, cin, 1, 1)
    # other strided shapes: MIOpen's channels-last weight-gradient kernel on channels-last operands (as NCHW tensors it copies
    # gy and transposes it back)
    return torch.ops.aten.convolution_backward(gy.contiguous(memory_format=torch.channels_last),
                                               x.contiguous(memory_format=torch.channels_last),
                                               torch.empty(wshape, dtype=x.dtype, device=x.device),
                                               None, [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1,
                                               [False, True, False])[1]


def _bn_tracking(bn):
    """(running_mean, running_var, momentum, num_batches_tracked) as nw_bn_relu_nhwc_train_fwd_f32 takes them."""
    if not (bn.track_running_stats and bn.running_mean is not None):
        return None, None, 0.0, None
    if bn.momentum is None:
        bn.num_batches_tracked += 1
        return bn.running_mean, bn.running_var, 1.0 / float(bn.num_batches_tracked), None
    return bn.running_mean, bn.running_var, float(bn.momentum), bn.num_batches_tracked


# BatchNorm's backward sums from the data-gradient convolutions' epilogues (nw_conv2d_nhwc_bnstat_f16x2) instead of a statistics
# pass: one launch and one read of (x, dy) less per BatchNorm, paid for by the x loads of the epilogue.  Measured on K4, three
# alternations on one box: 18.93-18.97 ms with, 18.79-18.83 without -- left OFF
DENSE_BWD_STATS_IN_DGRAD = os.environ.get("NW_DENSE_BWD_STATS", "0") == "1"
# norm1 -> relu1 -> conv1 backward as nw_bn_dgrad1x1_bwd_f16x2 (two streaming passes, no (rows, c) gradient tensor) instead of the
# data-gradient convolution + the two BatchNorm backward passes.  Half the HBM bytes, but measured SLOWER on K4 (15.3 vs 14.5 ms,
# alternated on one box): its first pass is instruction-bound (the per-channel sums) and the 14 x 14 / 7 x 7 layers pay three
# launch-latency floors either way -- left OFF (DESIGN.md 4.7h')
DENSE_FUSED_NORM1_BWD = os.environ.get("NW_DENSE_FUSED_NORM1_BWD", "0") == "1"


class _DenseBlockNhwcFn(torch.autograd.Function):
    """A whole dense block (model/densenet.py:62-80: every layer reads the concatenation of the block's input and all
    earlier layers' outputs) as ONE autograd node over ONE slab: the block's output tensor (rows = n h w, C0 + L growth
    floats per row) is allocated once, the input is copied into its first C0 channels, and every layer's 3x3 convolution
    writes its `growth` channels into place (row stride = the slab's width), so no concatenation is ever made; norm1 of a
    layer reads a channel prefix of the slab.  The backward keeps ONE gradient slab: a layer's gradient is a channel window
    of it, and norm1's backward kernel adds its dx into the prefix in place -- the O(L^2) gradient accumulations and slice
    copies autograd derives for the concatenations do not exist.  Same kernels and arithmetic as the layer-by-layer path
    (ops.bn_relu_train_nhwc, ops.conv2d_nhwc_train); parameter gradients in the order of `params`:
    per layer norm1.weight, norm1.bias, conv1.weight, norm2.weight, norm2.bias, conv2.weight."""

    @staticmethod
    def forward(ctx, x, layers, bank, *params):
        lib = _lib.load()
        xv, ldx0 = _nhwc_rows(x.detach())
        n, c0, h, w = xv.shape
        rows, dev = n * h * w, xv.device
        L = len(layers)
        growth = layers[0].conv2.weight.shape[0]
        mid = layers[0].conv1.weight.shape[0]
        ctot = c0 + L * growth
        slab = getattr(x, "nw_slab", None)            # the producer left room behind x's channels (_with_room): no copy
        if not (slab is not None and tuple(slab.shape) == (n, ctot, h, w) and slab.data_ptr() == xv.data_ptr() and ldx0 == ctot
                and slab.dtype == torch.float32 and slab.is_contiguous(memory_format=torch.channels_last)):
            slab = torch.empty((n, ctot, h, w), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
            slab[:, :c0] = xv
        st = _stream(xv)
        saved, meta = [], []
        f32 = dict(dtype=torch.float32, device=dev)
        eps = float(layers[0].norm1.eps)
        # Batch statistics per slab channel, computed ONCE: a channel's mean and variance are the same for every later layer's
        # norm1, the block's input channels get one pass here, and each convolution leaves the moments of the channels it writes
        # (nw_conv2d_nhwc_f16x2's `moments`), so no BatchNorm of the block reads its input for statistics.
        bstat = torch.empty(5 * ctot, **f32)                       # mean | invstd | var | min | max of the slab's channels
        bm, bi, bv, blo, bhi = (bstat[k * ctot:(k + 1) * ctot] for k in range(5))
        with _OnDevice(dev):
            wsb = lib.nw_bn_nhwc_minmax_workspace_bytes(rows, c0)
            ws = _workspace(wsb, dev)
            _lib.check(lib.nw_bn_nhwc_moments_minmax_f32(_ptr(slab), ctot, rows, c0, eps, _ptr(bm), _ptr(bi), _ptr(bv), _ptr(blo),
                                                         _ptr(bhi), _ptr(ws), wsb, st), "nw_bn_nhwc_moments_minmax_f32")
            def layer_params(k):
                g1, b1, w1, g2, b2, w2 = (t.detach() for t in params[6 * k:6 * k + 6])
                return _f32c(g1), _f32c(b1), w1, _f32c(g2), _f32c(b2), w2
            # Round 4: neither t1 = relu(norm1(prefix)) nor t2 = relu(norm2(u)) exists: the convolutions' loaders apply the
            # BatchNorm + ReLU on the way into LDS (nw_conv2d_nhwc_bnrelu_f16x2) from a per-channel table mean | a | beta, with
            # the exact bound on the transformed tensor (from each channel's minimum and maximum, which the producing
            # convolution's epilogue leaves beside the moments) as its amax record.  The table of a layer's norm1 is written by
            # the launch that merges the previous layer's fresh channels (nw_bn_nhwc_prep_window_from_partials_f32): four
            # launches per layer -- conv1, merge + norm2's table, conv2, merge + the next norm1's table.
            g1, b1, w1, g2, b2, w2 = layer_params(0)
            tab1 = torch.empty(3 * c0, **f32)
            am1 = torch.empty(AMAX_SLOTS, **f32)
            rm, rv, mom, nbt = _bn_tracking(layers[0].norm1)
            _lib.check(lib.nw_bn_nhwc_prep_f32(_ptr(bm), _ptr(bi), _ptr(bv), _ptr(blo), _ptr(bhi), _ptr(g1), _ptr(b1), _ptr(rm), _ptr(rv),
                                               _ptr(nbt), mom, 1, rows, c0, _ptr(tab1), _ptr(am1), st), "nw_bn_nhwc_prep_f32")
            for k, layer in enumerate(layers):
                c = c0 + k * growth
                o1, o2 = bank.operands(layer.conv1.weight), bank.operands(layer.conv2.weight)
                kh = w2.shape[2]
                u = torch.empty((rows, mid), **f32)
                stats = torch.empty(5 * mid, **f32)                        # norm2: mean | invstd | var | min | max
                tab2 = torch.empty(3 * mid, **f32)
                am2 = torch.empty(AMAX_SLOTS, **f32)                       # amax record of relu(norm2(u)) (am1: of relu(norm1(prefix)))
                m2, i2, v2, lo2, hi2 = (stats[j * mid:(j + 1) * mid] for j in range(5))
                G1 = lib.nw_conv2d_nhwc_moments_groups(n, h, w, c, mid, 1, 1, 1, 0)
                G2 = lib.nw_conv2d_nhwc_moments_groups(n, h, w, mid, growth, kh, kh, 1, kh // 2)
                part = _workspace(4 * 5 * max(G1 * mid, G2 * growth), dev)
                _lib.check(lib.nw_conv2d_nhwc_bnrelu_f16x2(_ptr(slab), _ptr(tab1), _ptr(am1), 0, _ptr(o1[0].split), _ptr(o1[0].scale), None, 0,
                                                           _ptr(u), None, n, h, w, c, mid, 1, 1, 1, 0, ctot, 0, _ptr(part), st),
                           "nw_conv2d_nhwc_bnrelu_f16x2")
                rm, rv, mom, nbt = _bn_tracking(layer.norm2)
                _lib.check(lib.nw_bn_nhwc_prep_from_partials_f32(_ptr(part), G1, mid, eps, _ptr(m2), _ptr(i2), _ptr(v2), _ptr(lo2),
                                                                 _ptr(hi2), _ptr(g2), _ptr(b2), _ptr(rm), _ptr(rv), _ptr(nbt), mom, 1,
                                                                 _ptr(tab2), _ptr(am2), st), "nw_bn_nhwc_prep_from_partials_f32")
                _lib.check(lib.nw_conv2d_nhwc_bnrelu_f16x2(_ptr(u), _ptr(tab2), _ptr(am2), 0, _ptr(o2[0].split), _ptr(o2[0].scale), None, 0,
                                                           slab.data_ptr() + 4 * c, None, n, h, w, mid, growth, kh, kh, 1, kh // 2, 0,
                                                           ctot, _ptr(part), st), "nw_conv2d_nhwc_bnrelu_f16x2")
                saved += [u, stats, tab1, am1, tab2, am2, g1, b1, g2, b2]
                meta.append((c, kh, o1[1], o2[1], tuple(w1.shape), tuple(w2.shape)))
                if k + 1 < L:      # the fresh channels' statistics + the whole table of the next layer's norm1
                    g1, b1, w1, g2, b2, w2 = layer_params(k + 1)
                    tab1 = torch.empty(3 * (c + growth), **f32)
                    am1 = torch.empty(AMAX_SLOTS, **f32)
                    rm, rv, mom, nbt = _bn_tracking(layers[k + 1].norm1)
                    _lib.check(lib.nw_bn_nhwc_prep_window_from_partials_f32(
                        _ptr(part), G2, growth, c, c, rows, eps, _ptr(bm), _ptr(bi), _ptr(bv), _ptr(blo), _ptr(bhi), _ptr(g1), _ptr(b1),
                        _ptr(rm), _ptr(rv), _ptr(nbt), mom, 1, _ptr(tab1), _ptr(am1), st), "nw_bn_nhwc_prep_window_from_partials_f32")
                else:
                    _lib.check(lib.nw_bn_nhwc_prep_window_from_partials_f32(
                        _ptr(part), G2, growth, c, 0, rows, eps, _ptr(bm), _ptr(bi), _ptr(bv), _ptr(blo), _ptr(bhi), None, None, None,
                        None, None, 0.0, 1, None, None, st), "nw_bn_nhwc_prep_window_from_partials_f32")
        saved.append(bstat)
        ctx.save_for_backward(slab, *saved)
        ctx.meta, ctx.dims = meta, (n, c0, h, w, growth, mid, ctot)
        return slab

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        lib = _lib.load()
        slab, *saved = ctx.saved_tensors
        bstat = saved[-1]
        n, c0, h, w, growth, mid, ctot = ctx.dims
        rows, dev = n * h * w, slab.device
        am_g = getattr(gout, "nw_amax", None)
        G = gout.float().contiguous(memory_format=torch.channels_last)
        if G.data_ptr() == gout.data_ptr() and not getattr(gout, "nw_fresh", False):
            G = G.clone(memory_format=torch.channels_last)             # the gradient slab is updated in place: a buffer of
                                                                       #   our own BatchNorm backward (`nw_fresh`) is adopted
        if am_g is None:
            am_g = absmax(G)
        st = _stream(slab)
        f32 = dict(dtype=torch.float32, device=dev)
        grads = [None] * (6 * len(ctx.meta))
        # the weight gradients are not on the critical path: collected here and run together behind the loop
        # (nw_conv2d_nhwc_wgrad_batch_f16x2: a dozen workgroups per 14x14 / 7x7 layer fill the chip only together)
        wjobs, wkeep = [], []
        with _OnDevice(dev):
            for k in range(len(ctx.meta) - 1, -1, -1):
                c, kh, d1, d2, w1s, w2s = ctx.meta[k]
                u, stats, tab1, am1, tab2, am2, g1, b1, g2, b2 = saved[10 * k:10 * k + 10]
                m1, i1, m2, i2 = bstat[:c], bstat[ctot:ctot + c], stats[:mid], stats[mid:2 * mid]
                # (tab1 / tab2: the forward's tables -- the weight gradients' x operands are relu(norm(.)) again, applied by
                #  THEIR loaders)
                gv = G.data_ptr() + 4 * c                              # this layer's window of the gradient slab
                # conv2 (3x3): weight gradient, data gradient
                dw2 = torch.empty((growth, mid, kh, kh), **f32)          # torch's layout: autograd takes it without a copy
                wjobs.append(_lib.WgradJob(_ptr(u), _ptr(am2), gv, _ptr(am_g), _ptr(dw2),
                                           n, h, w, mid, growth, kh, kh, 1, kh // 2, 0, ctot, 1, _ptr(tab2)))
                wkeep.append(am_g)
                if DENSE_BWD_STATS_IN_DGRAD:
                    dt2 = torch.empty((rows, mid), **f32)
                    am_d = torch.empty(3 * AMAX_SLOTS, **f32)              # amax records of dt2 | du | dt1
                    # the data gradients also leave BatchNorm's backward sums of what they write (sum g, sum g xhat per pixel
                    # group), so the BatchNorm backward is the sum of the groups + ONE pass (dx), not two
                    Gd2 = lib.nw_conv2d_nhwc_moments_groups(n, h, w, growth, mid, kh, kh, 1, kh - 1 - kh // 2)
                    Gd1 = lib.nw_conv2d_nhwc_moments_groups(n, h, w, mid, c, 1, 1, 1, 0)
                    bpart = torch.empty(2 * max(Gd2 * mid, Gd1 * c), **f32)
                    bs2 = _lib.ConvBnStat(_ptr(u), mid, _ptr(m2), _ptr(i2), _ptr(g2), _ptr(b2), _ptr(bpart))
                    _lib.check(lib.nw_conv2d_nhwc_bnstat_f16x2(gv, _ptr(am_g), _ptr(d2.split), _ptr(d2.scale), _ptr(dt2), _ptr(am_d),
                                                               n, h, w, growth, mid, kh, kh, 1, kh - 1 - kh // 2, ctot, 0,
                                                               ctypes.byref(bs2), st), "nw_conv2d_nhwc_bnstat_f16x2")
                    # norm2 + relu
                    du = torch.empty((rows, mid), **f32)
                    dg2, db2 = torch.empty(mid, **f32), torch.empty(mid, **f32)
                    kws = torch.empty(2 * max(c, mid), **f32)
                    _lib.check(lib.nw_bn_relu_nhwc_train_bwd_from_partials_f32(
                        _ptr(u), mid, _ptr(dt2), _ptr(g2), _ptr(b2), _ptr(m2), _ptr(i2), _ptr(bpart), Gd2, _ptr(du), _ptr(dg2), _ptr(db2),
                        None, 0, 0, am_d.data_ptr() + 4 * AMAX_SLOTS, _ptr(kws), 4 * kws.numel(), rows, mid, st),
                        "nw_bn_relu_nhwc_train_bwd_from_partials_f32")
                    # conv1 (1x1)
                    dw1 = torch.empty((mid, c, 1, 1), **f32)     # (a 1x1 kernel: (Cout, 1, 1, Cin) bytes ARE torch's (Cout, Cin, 1, 1): no copy in AccumulateGrad)
                    wjobs.append(_lib.WgradJob(_ptr(slab), _ptr(am1), _ptr(du), am_d.data_ptr() + 4 * AMAX_SLOTS, _ptr(dw1),
                                               n, h, w, c, mid, 1, 1, 1, 0, ctot, 0, 0, _ptr(tab1)))
                    wkeep += [du, am_d]
                    dt1 = torch.empty((rows, c), **f32)
                    bs1 = _lib.ConvBnStat(_ptr(slab), ctot, _ptr(m1), _ptr(i1), _ptr(g1), _ptr(b1), _ptr(bpart))
                    _lib.check(lib.nw_conv2d_nhwc_bnstat_f16x2(_ptr(du), am_d.data_ptr() + 4 * AMAX_SLOTS, _ptr(d1.split), _ptr(d1.scale),
                                                               _ptr(dt1), am_d.data_ptr() + 8 * AMAX_SLOTS, n, h, w, mid, c, 1, 1, 1, 0, 0, 0,
                                                               ctypes.byref(bs1), st), "nw_conv2d_nhwc_bnstat_f16x2")
                    # norm1 + relu over the slab's prefix: dx is ADDED into the gradient slab's prefix, in place
                    dg1, db1 = torch.empty(c, **f32), torch.empty(c, **f32)
                    am_g = torch.empty(AMAX_SLOTS, **f32)
                    _lib.check(lib.nw_bn_relu_nhwc_train_bwd_from_partials_f32(
                        _ptr(slab), ctot, _ptr(dt1), _ptr(g1), _ptr(b1), _ptr(m1), _ptr(i1), _ptr(bpart), Gd1, _ptr(G), _ptr(dg1), _ptr(db1),
                        _ptr(G), ctot, ctot, _ptr(am_g), _ptr(kws), 4 * kws.numel(), rows, c, st),
                        "nw_bn_relu_nhwc_train_bwd_from_partials_f32")
                else:
                    dt2 = torch.empty((rows, mid), **f32)
                    am_d = torch.empty(3 * AMAX_SLOTS, **f32)              # amax records of dt2 | du | dt1
                    _lib.check(lib.nw_conv2d_nhwc_f16x2(gv, _ptr(am_g), _ptr(d2.split), _ptr(d2.scale), None, None, 0, _ptr(dt2),
                                                        _ptr(am_d), n, h, w, growth, mid, kh, kh, 1, kh - 1 - kh // 2, ctot, 0, None, st),
                               "nw_conv2d_nhwc_f16x2")
                    # norm2 + relu
                    du = torch.empty((rows, mid), **f32)
                    dg2, db2 = torch.empty(mid, **f32), torch.empty(mid, **f32)
                    bnb = max(lib.nw_bn_nhwc_workspace_bytes(rows, c), lib.nw_bn_nhwc_workspace_bytes(rows, mid))
                    wsn = _workspace(bnb, dev)
                    _lib.check(lib.nw_bn_relu_nhwc_train_bwd_f32(_ptr(u), mid, _ptr(dt2), _ptr(g2), _ptr(b2), _ptr(m2), _ptr(i2),
                                                                 _ptr(du), _ptr(dg2), _ptr(db2), None, 0, 0,
                                                                 am_d.data_ptr() + 4 * AMAX_SLOTS, _ptr(wsn), bnb, rows, mid, 1, st),
                               "nw_bn_relu_nhwc_train_bwd_f32")
                    # conv1 (1x1)
                    dw1 = torch.empty((mid, c, 1, 1), **f32)     # (a 1x1 kernel: (Cout, 1, 1, Cin) bytes ARE torch's (Cout, Cin, 1, 1): no copy in AccumulateGrad)
                    wjobs.append(_lib.WgradJob(_ptr(slab), _ptr(am1), _ptr(du), am_d.data_ptr() + 4 * AMAX_SLOTS, _ptr(dw1),
                                               n, h, w, c, mid, 1, 1, 1, 0, ctot, 0, 0, _ptr(tab1)))
                    wkeep += [du, am_d]
                    dg1, db1 = torch.empty(c, **f32), torch.empty(c, **f32)
                    am_g = torch.empty(AMAX_SLOTS, **f32)
                    if DENSE_FUSED_NORM1_BWD and mid % 32 == 0 and mid <= 128 and c % 32 == 0:
                        # conv1's data gradient, norm1's statistics and dx in two streaming passes: the (rows, c) gradient never
                        # reaches memory (csrc/bn_dgrad.hip); dx is ADDED into the gradient slab's prefix, in place
                        fb = lib.nw_bn_dgrad1x1_workspace_bytes(rows, c)
                        wsf = _workspace(fb, dev)
                        _lib.check(lib.nw_bn_dgrad1x1_bwd_f16x2(_ptr(du), am_d.data_ptr() + 4 * AMAX_SLOTS, _ptr(d1.split), _ptr(d1.scale),
                                                                _ptr(slab), ctot, _ptr(tab1), c, _ptr(i1), _ptr(G), ctot, _ptr(am_g),
                                                                _ptr(dg1), _ptr(db1), _ptr(wsf), fb, rows, c, mid, st),
                                   "nw_bn_dgrad1x1_bwd_f16x2")
                    else:
                        dt1 = torch.empty((rows, c), **f32)
                        _lib.check(lib.nw_conv2d_nhwc_f16x2(_ptr(du), am_d.data_ptr() + 4 * AMAX_SLOTS, _ptr(d1.split), _ptr(d1.scale), None,
                                                            None, 0, _ptr(dt1), am_d.data_ptr() + 8 * AMAX_SLOTS, n, h, w, mid, c, 1, 1, 1, 0,
                                                            0, 0, None, st), "nw_conv2d_nhwc_f16x2")
                        # norm1 + relu over the slab's prefix: dx is ADDED into the gradient slab's prefix, in place
                        _lib.check(lib.nw_bn_relu_nhwc_train_bwd_f32(_ptr(slab), ctot, _ptr(dt1), _ptr(g1), _ptr(b1), _ptr(m1), _ptr(i1),
                                                                     _ptr(G), _ptr(dg1), _ptr(db1), _ptr(G), ctot, ctot, _ptr(am_g),
                                                                     _ptr(wsn), bnb, rows, c, 1, st), "nw_bn_relu_nhwc_train_bwd_f32")
                grads[6 * k:6 * k + 6] = [dg1, db1, dw1, dg2, db2, dw2]
            if wjobs:
                jobs = (_lib.WgradJob * len(wjobs))(*wjobs)
                wsb = lib.nw_conv2d_nhwc_wgrad_batch_workspace_bytes(jobs, len(wjobs))
                ws = _workspace(wsb, dev)
                _lib.check(lib.nw_conv2d_nhwc_wgrad_batch_f16x2(jobs, len(wjobs), _ptr(ws), wsb, st),
                           "nw_conv2d_nhwc_wgrad_batch_f16x2")
        del wkeep
        dx = None
        if ctx.needs_input_grad[0]:
            dx = G[:, :c0]                  # a channel prefix of the gradient slab (the pools' backward reads strided rows)
            dx.nw_amax = am_g
        return (dx, None, None, *grads)


@torch.no_grad()
def bn_relu_nhwc_apply(x, mean, invstd, gamma, beta, relu=True):
    """relu((x - mean) gamma invstd + beta) over a channels-last fp32 activation with GIVEN statistics (an eval-mode BatchNorm2d
    + ReLU that no convolution follows: DenseNet's norm5) in nw_bn_relu_nhwc_apply_f32; (n, c, h, w), c % 4 == 0."""
    _need_hip(x)
    lib = _lib.load()
    xv, ldx = _nhwc_rows(x)
    n, c, h, w = xv.shape
    y = torch.empty((n, c, h, w), dtype=torch.float32, device=xv.device, memory_format=torch.channels_last)
    amax = torch.empty(AMAX_SLOTS, dtype=torch.float32, device=xv.device)
    with _OnDevice(xv.device):
        _lib.check(lib.nw_bn_relu_nhwc_apply_f32(_ptr(xv), ldx, _ptr(mean), _ptr(invstd), None, _ptr(gamma), _ptr(beta), None, None, None,
                                                 0.0, _ptr(y), _ptr(amax), n * h * w, c, int(relu), _stream(xv)), "nw_bn_relu_nhwc_apply_f32")
    y.nw_amax = amax
    return y


@torch.no_grad()
def bn_relu_avgpool2_nhwc(x, tab):
    """avgpool2x2(relu((x - mean) a + beta)) over a channels-last fp32 activation (tab = bn_table(bn)): an eval-mode DenseNet
    transition's norm -> relu with the pool pulled in front of the 1 x 1 convolution (nw_bn_relu_avgpool2x2_nhwc_f32)."""
    _need_hip(x, tab)
    lib = _lib.load()
    xv, ldx = _nhwc_rows(x)
    n, c, h, w = xv.shape
    y = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=xv.device, memory_format=torch.channels_last)
    amax = torch.empty(AMAX_SLOTS, dtype=torch.float32, device=xv.device)
    with _OnDevice(xv.device):
        _lib.check(lib.nw_bn_relu_avgpool2x2_nhwc_f32(_ptr(xv), ldx, _ptr(tab), _ptr(y), 0, _ptr(amax), n, h, w, c, _stream(xv)),
                   "nw_bn_relu_avgpool2x2_nhwc_f32")
    y.nw_amax = amax
    return y


def bn_table(bn):
    """mean | a | beta (3 c floats) of an eval-mode BatchNorm2d for nw_conv2d_nhwc_bnrelu_f16x2: y = (x - mean) a + beta."""
    a = bn.weight.detach().float() * torch.rsqrt(bn.running_var.detach().float() + bn.eps)
    return torch.cat((bn.running_mean.detach().float(), a, bn.bias.detach().float())).contiguous()


@torch.no_grad()
def dense_block_nhwc_infer(x, plan):
    """A dense block (model/densenet.py:62-80) at inference over ONE channels-last slab, two launches per layer:
    conv1 = nw_conv2d_nhwc_bnrelu_f16x2 (norm1 + relu1 in its loaders, norm2 folded into its weight and bias, relu2 in its
    store) and conv2 = nw_conv2d_nhwc_f16x2 writing its `growth` channels into the slab.  plan: per layer (table of norm1,
    SplitConvWeight of the folded conv1, its bias, SplitConvWeight of conv2).  The bound conv1's loaders need is derived in the
    kernel from the amax records of everything in the slab so far (the input's and every conv2's: `raw_records`).  Returns the
    slab (n, C0 + L growth, h, w) with `.nw_records` (L + 1 records)."""
    lib = _lib.load()
    xv, ldx0 = _nhwc_rows(x)
    n, c0, h, w = xv.shape
    dev, L = xv.device, len(plan)
    growth, mid = plan[0][3].shape[0], plan[0][1].shape[0]
    ctot = c0 + L * growth
    slab = getattr(x, "nw_slab", None)
    if not (slab is not None and tuple(slab.shape) == (n, ctot, h, w) and slab.data_ptr() == xv.data_ptr() and ldx0 == ctot
            and slab.dtype == torch.float32 and slab.is_contiguous(memory_format=torch.channels_last)):
        slab = torch.empty((n, ctot, h, w), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
        slab[:, :c0] = xv
    f32 = dict(dtype=torch.float32, device=dev)
    recs = torch.empty((L + 1) * AMAX_SLOTS, **f32)
    am0 = getattr(x, "nw_amax", None)
    recs[:AMAX_SLOTS].copy_(am0 if am0 is not None else absmax(xv if ldx0 == c0 else xv.contiguous(memory_format=torch.channels_last)))
    u = torch.empty((n * h * w, mid), **f32)
    am_u = torch.empty(AMAX_SLOTS, **f32)
    st = _stream(xv)
    with _OnDevice(dev):
        for k, (tab1, w1, b1, w2) in enumerate(plan):
            c = c0 + k * growth
            kh = w2.shape[2]
            _lib.check(lib.nw_conv2d_nhwc_bnrelu_f16x2(_ptr(slab), _ptr(tab1), _ptr(recs), k + 1, _ptr(w1.split), _ptr(w1.scale), _ptr(b1),
                                                       1, _ptr(u), _ptr(am_u), n, h, w, c, mid, 1, 1, 1, 0, ctot, 0, None, st),
                       "nw_conv2d_nhwc_bnrelu_f16x2")
            _lib.check(lib.nw_conv2d_nhwc_f16x2(_ptr(u), _ptr(am_u), _ptr(w2.split), _ptr(w2.scale), None, None, 0,
                                                slab.data_ptr() + 4 * c, recs.data_ptr() + 4 * AMAX_SLOTS * (k + 1), n, h, w, mid, growth,
                                                kh, kh, 1, kh // 2, 0, ctot, None, st), "nw_conv2d_nhwc_f16x2")
    slab.nw_records = recs
    return slab


@torch.no_grad()
def conv1x1_bnrelu_nhwc_infer(x, tab, weight, records=None, room=0):
    """conv1x1(relu(bn(x))) of a DenseNet transition at inference (model/densenet.py:86-90; nw_conv2d_nhwc_bnrelu_f16x2)
    over a channels-last activation; records: the amax records of x's producers (default: x.nw_records / x.nw_amax)."""
    lib = _lib.load()
    xv, ldx = _nhwc_rows(x)
    n, c, h, w = xv.shape
    cout = weight.shape[0]
    if records is None:
        records = getattr(x, "nw_records", None)
        if records is None:
            records = getattr(x, "nw_amax", None)
        if records is None:
            records = absmax(xv if ldx == c else xv.contiguous(memory_format=torch.channels_last))
    y, ldy = _with_room(n, cout, h, w, room, xv.device)
    am = torch.empty(AMAX_SLOTS, dtype=torch.float32, device=xv.device)
    with _OnDevice(xv.device):
        _lib.check(lib.nw_conv2d_nhwc_bnrelu_f16x2(_ptr(xv), _ptr(tab), _ptr(records), records.numel() // AMAX_SLOTS, _ptr(weight.split),
                                                   _ptr(weight.scale), None, 0, _ptr(y), _ptr(am), n, h, w, c, cout, 1, 1, 1, 0, ldx,
                                                   ldy if room > 0 else 0, None, _stream(xv)), "nw_conv2d_nhwc_bnrelu_f16x2")
    y.nw_amax = am
    return y


def dense_block_nhwc_supported(x, layers, bank):
    """Can ops.dense_block_nhwc_train run this block?  (channels-last fp32 on the device, a weight bank that holds every
    layer's forward and data-gradient operands, affine BatchNorms, no dropout, shapes the convolution kernels serve)"""
    if bank is None or not x.is_cuda or x.dim() != 4 or not layers:
        return False
    lib = _lib.load()
    n, c0, h, w = x.shape
    growth, mid = layers[0].conv2.weight.shape[0], layers[0].conv1.weight.shape[0]
    if n * h * w <= 1 or c0 % 32 or growth % 32 or mid % 32 or c0 + len(layers) * growth > BN_NHWC_MAX_C:
        return False
    for k, layer in enumerate(layers):
        c = c0 + k * growth
        w1, w2 = layer.conv1.weight, layer.conv2.weight
        kh = w2.shape[2]
        if (layer.drop_rate > 0 or tuple(w1.shape) != (mid, c, 1, 1) or tuple(w2.shape) != (growth, mid, kh, kh) or kh % 2 == 0
                or not (layer.norm1.affine and layer.norm2.affine) or layer.norm1.weight is None or layer.norm2.weight is None
                or layer.norm1.eps != layers[0].norm1.eps or layer.norm2.eps != layers[0].norm1.eps):
            return False
        for wt in (w1, w2):
            ops_ = bank.operands(wt) if bank.has(wt) else None
            if ops_ is None or ops_[0] is None or ops_[1] is None:
                return False
        if not (lib.nw_conv2d_nhwc_wgrad_supported(n, h, w, c, mid, 1, 1, 1, 0)
                and lib.nw_conv2d_nhwc_wgrad_supported(n, h, w, mid, growth, kh, kh, 1, kh // 2)):
            return False
    return True


def dense_block_nhwc_train(x, layers, bank):
    """The dense block `layers` (modules with norm1, conv1, norm2, conv2) over a channels-last activation as one autograd node
    over one slab (_DenseBlockNhwcFn); returns the block's output (n, C0 + L growth, h, w), channels_last."""
    params = []
    for layer in layers:
        params += [layer.norm1.weight, layer.norm1.bias, layer.conv1.weight, layer.norm2.weight, layer.norm2.bias,
                   layer.conv2.weight]
    _need_hip(x, *params)
    return _DenseBlockNhwcFn.apply(x, tuple(layers), bank, *params)


def conv2d_nhwc_train(x, weight, stride=1, pad=0, amax=None, operands=None, room=0):
    """Differentiable conv2d(x, weight, stride=stride, padding=pad) for channels-last fp32 activations on the MI355X
    (bias-free: the backbones' convolutions have none).  x may carry `.nw_amax`; operands: ConvWeightBank.operands(weight)
    (else the weight is split inside the call, forward and backward)."""
    _need_hip(x, weight)
    return _ConvNhwcFn.apply(x, weight, int(stride), int(pad), amax, operands, int(room))
