"""'full' inference against a support bank sharded over the ranks of one node.

Rank g owns a contiguous slice of the (class-sorted) bank.  Queries are replicated; each rank runs
the partial forward over its slice (HIP), the ranks exchange ONE packed buffer per bucket of query
batches -- [m | den | num] per batch -- with a single RCCL all-gather over xGMI (payloads are a few
hundred KB: latency-bound, so batches are bucketed and the collective of bucket i overlaps the
kernels of bucket i+1), and every rank merges to the same (B,C) log-probabilities.

    one process per GPU, torch.distributed backend "nccl" (= RCCL on ROCm); "gloo" in CPU tests,
    where the compute hooks are replaced by the oracle (tests/test_sharded_gloo.py).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import ops


def shard_bounds(n_rows: int, world: int, rank: int):
    """Contiguous, near-equal split: rows [lo, hi) of the bank belong to `rank`."""
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _coalesce(chunk):
    """One (nb*B, d) tensor for a bucket of (B, d) query batches: a VIEW when the batches already sit back to
    back in one storage (slices of a staging buffer, the usual case in a serving loop), else a copy."""
    first = chunk[0]
    if first.is_contiguous() and first.dim() == 2:
        base = first.untyped_storage().data_ptr()
        step = first.numel() * first.element_size()
        if all(c.dtype == first.dtype and c.shape == first.shape and c.is_contiguous() and
               c.untyped_storage().data_ptr() == base and c.data_ptr() == first.data_ptr() + k * step
               for k, c in enumerate(chunk)):
            return first.as_strided((len(chunk) * first.shape[0], first.shape[1]), (first.shape[1], 1))
    return torch.cat(chunk, dim=0)


class ShardedBank:
    def __init__(self, feat_shard, y_shard, n_classes, kind="euclidean", logit_scale=None, group=None,
                 partial_fn=None, merge_fn=None, persistent_wgs=None):
        """persistent_wgs: workgroups of the persistent tile kernel (nw_fwd_opts.persistent_wgs, a multiple of 8; 0 = one per
        CU).  Default: with more than one rank, all CUs but one per XCD (count - 8) -- the all-gather of bucket i runs
        under the kernels of bucket i + 1, and a kernel that holds one 160 KB-LDS workgroup on EVERY CU would leave RCCL's
        kernel nowhere to run until it ends; with one rank, one per CU."""
        self.feat = feat_shard.detach().to(torch.float32).contiguous()
        self.y = y_shard.detach().to(torch.int64).contiguous()
        self.C = int(n_classes)
        self.kind, self.logit_scale, self.group = kind, logit_scale, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._partial = partial_fn or self._hip_partial
        self._merge = merge_fn or self._hip_merge
        self._ws = None
        if persistent_wgs is None:
            persistent_wgs = 0
            if self.world > 1 and self.feat.is_cuda:
                cus = torch.cuda.get_device_properties(self.feat.device).multi_processor_count
                persistent_wgs = max(8, (cus - 8) // 8 * 8)
        self.persistent_wgs = int(persistent_wgs)
        # Class windows: a contiguous slice of a class-sorted bank holds only ~C/G classes, so its
        # partial forward runs on labels shifted by the slice's lowest class with CL = widest window
        # over the ranks; the exchanged rows are (2 + CL) instead of (2 + C) floats per query.
        self.class_lo, self.CL, self.y_local = None, self.C, self.y
        if self.world > 1:
            lo = int(self.y.min()) if self.y.numel() else 0
            hi = int(self.y.max()) if self.y.numel() else -1
            box = torch.tensor([lo, hi], dtype=torch.int64, device=self.feat.device)
            allb = torch.empty(self.world, 2, dtype=torch.int64, device=self.feat.device)
            dist.all_gather_into_tensor(allb.view(-1), box, group=group)
            width = int((allb[:, 1] - allb[:, 0] + 1).clamp_min(1).max())
            if width < self.C:
                self.CL = width
                self.class_lo = allb[:, 0].clamp(0, max(self.C - 1, 0)).contiguous()
                self.y_local = (self.y - lo).contiguous()
        # the shard never changes: prepare it once (squared norms + split-fp16 rows, ops.SplitBank)
        self.cache = ops.SplitBank(self.feat) if (partial_fn is None and self.feat.is_cuda) else None
        self.norm2 = self.cache.norm2 if self.cache is not None else None
        if self.cache is not None and self.cache.split is not None:
            self.cache.build_tables(self.y_local)   # the labels every call of this shard passes (self.y when there is one rank)

    # ---- HIP compute hooks (the product path)
    def _hip_partial(self, packed_row, q):
        N, d = self.feat.shape
        if self.cache is not None and self.cache.pad:
            d = self.cache.shape[1]              # a shard of a width that is not a multiple of 32: the bank's padded rows
        B = q.shape[0]
        need = ops._lib.load().nw_fwd_workspace_bytes(B, N, d, self.CL)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(need, 1), dtype=torch.uint8, device=q.device)
        ops.nw_partials_into(packed_row, q, self.feat, self.y_local, self.CL, self.kind, self.logit_scale,
                             ws=self._ws, cache=self.cache, persistent_wgs=self.persistent_wgs)

    def _hip_merge(self, gathered_rows, B):
        return ops.nw_merge(gathered_rows, B, self.C, class_lo=self.class_lo, c_local=self.CL)

    def row_len(self, B):
        return 2 * B + B * self.CL

    def predict(self, q):
        """One query batch: (B,d) -> (B,C) log-probabilities, identical on every rank."""
        return self.predict_stream([q], bucket=1)[0]

    def predict_stream(self, batches, bucket=8):
        """Pipelined prediction of a list of (B_i, d) query batches (B_i may differ: the ragged tail of a
        loader is an ordinary case).

        Every `bucket` consecutive batches are coalesced into ONE launch of the partial forward (the
        kernel tiles over queries anyway, and one launch over the bucket's rows amortises the launch,
        the tile prologue and the merge), ONE packed buffer [m | den | num] and ONE all-gather; the
        merge of bucket i runs after the kernels of bucket i+1 have been queued, so the collective
        flies under them.  Outputs are returned per batch, in order."""
        if not batches:
            return []
        d = batches[0].shape[-1]
        for k, qb in enumerate(batches):
            if qb.dim() != 2 or qb.shape[1] != d:
                raise ValueError(f"predict_stream: batch {k} has shape {tuple(qb.shape)}, expected (B, {d})")
        dev, G = self.feat.device, self.world
        outs, pending = [], None
        ring = {}

        def split(out, sizes):
            pos = 0
            for n in sizes:
                outs.append(out[pos:pos + n])
                pos += n

        def finish(p):
            work, gathered, sizes = p
            if work is not None:
                work.wait()
            split(self._merge(gathered, sum(sizes)), sizes)          # (rows of the bucket, C)

        for n_bucket, i0 in enumerate(range(0, len(batches), bucket)):
            chunk = batches[i0:i0 + bucket]
            sizes = [int(c.shape[0]) for c in chunk]
            Bq = sum(sizes)
            L = self.row_len(Bq)
            key = (n_bucket % 3, Bq)
            if key not in ring:
                ring[key] = (torch.empty(L, dtype=torch.float32, device=dev),
                             torch.empty(G, L, dtype=torch.float32, device=dev))
            packed, gathered = ring[key]
            qcat = chunk[0] if len(chunk) == 1 else _coalesce(chunk)
            if G == 1 and self._partial == self._hip_partial:
                # one rank: nothing to exchange, the forward finalises in place
                split(ops.nw_head(qcat, self.feat, self.y, self.C, self.kind, self.logit_scale, support_cache=self.cache),
                      sizes)
                continue
            self._partial(packed, qcat.detach().to(torch.float32).contiguous())
            if G > 1:
                work = dist.all_gather_into_tensor(gathered.view(-1), packed, group=self.group, async_op=True)
            else:
                gathered, work = packed.view(1, L), None
            if pending is not None:
                finish(pending)
            pending = (work, gathered, sizes)
        if pending is not None:
            finish(pending)
        return outs
