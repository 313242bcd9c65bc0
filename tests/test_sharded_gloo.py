"""N>1 path on CPU: world_size-2 gloo, compute hooks replaced by the oracle (checker), the collective
plumbing (packing, bucketing, all-gather, merge order) is the product code in nwhead_amd/sharded.py."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from nwhead_amd.sharded import ShardedBank, shard_bounds
        from oracle import nw_oracle as O
        g = torch.Generator().manual_seed(11)
        B, N, d, C = 6, 103, 24, 5
        s = torch.randn(N, d, generator=g)
        sy = (torch.arange(N) % C).sort().values
        batches = [torch.randn(B, d, generator=g) for _ in range(4)] + [torch.randn(B - 2, d, generator=g)]   # ragged tail
        lo, hi = shard_bounds(N, world, rank)

        def partial_fn(row, qb):                      # oracle stands in for nw_fwd_partial_f32
            # (labels shifted into this rank's class window, CL classes: what the product path passes)
            m, den, num = O.nw_partials_f64(qb, s[lo:hi], bank.y_local, bank.CL)
            nq = qb.shape[0]                          # a bucket of batches arrives coalesced
            row[:nq] = m.float()
            row[nq:2 * nq] = den.float()
            row[2 * nq:] = num.float().reshape(-1)

        def merge_fn(rows, Bq):                       # oracle stands in for nw_merge_finalize_f32
            G, CL = rows.shape[0], bank.CL
            ms = [rows[k, :Bq].double() for k in range(G)]
            dens = [rows[k, Bq:2 * Bq].double() for k in range(G)]
            nums = []
            for k in range(G):                        # scatter every shard's window back to C classes
                full = torch.zeros(Bq, C, dtype=torch.float64)
                lo_k = int(bank.class_lo[k]) if bank.class_lo is not None else 0
                win = rows[k, 2 * Bq:2 * Bq + Bq * CL].double().reshape(Bq, CL)
                full[:, lo_k:lo_k + CL] = win[:, :C - lo_k]
                nums.append(full)
            return O.nw_merge_f64(ms, dens, nums).float()

        bank = ShardedBank(s[lo:hi], sy[lo:hi], C, partial_fn=partial_fn, merge_fn=merge_fn)
        outs = bank.predict_stream(batches, bucket=2)          # buckets of 2, 2, 1
        outs3 = bank.predict_stream(batches, bucket=3)         # 3, then 2 with the short batch inside the bucket
        assert [tuple(o.shape) for o in outs] == [(len(qb), C) for qb in batches]
        assert all(torch.equal(a, b) for a, b in zip(outs, outs3))
        try:
            bank.predict_stream([batches[0], torch.zeros(B, d + 1)])
            raise SystemExit("a batch with another feature width must be refused")
        except ValueError:
            pass
        one = bank.predict(batches[0])
        ref = [O.nw_head_f64(qb, s, sy, C).float() for qb in batches]
        err = max((o - r).abs().max().item() for o, r in zip(outs, ref))
        err = max(err, (one - ref[0]).abs().max().item())
        assert bank.CL < C and bank.class_lo is not None      # 2 ranks x 5 sorted classes: windows of 3
        q.put((rank, len(outs), err))
    finally:
        dist.destroy_process_group()


def test_sharded_predict_stream_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n, err in res:
        assert n == 5 and err < 2e-5, (rank, n, err)


# ------------------------------------------------------------------ NWNet.precompute_sharded
class _FakeImages(torch.utils.data.Dataset):
    """10 classes x 12 images of 3x4x4, labels interleaved (so the balanced bank has to re-order them)."""

    def __init__(self):
        g = torch.Generator().manual_seed(21)
        self.x = torch.randn(120, 3, 4, 4, generator=g)
        self.targets = [i % 10 for i in range(120)]

    def __len__(self):
        return len(self.targets)

    def __getitem__(self, i):
        return self.x[i], self.targets[i]


def _net():
    from nwhead_amd.nwhead.nw import NWNet
    torch.manual_seed(5)
    feat = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(48, 16))
    return NWNet(feat, 10, support_dataset=_FakeImages(), n_shot_full=7, device="cpu").eval()


def _worker_precompute(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from nwhead_amd.sharded import shard_bounds
        from oracle import nw_oracle as O
        net = _net()
        holder = {}

        def partial_fn(row, qb):
            bank = holder["bank"]
            m, den, num = O.nw_partials_f64(qb, bank.feat, bank.y_local, bank.CL)
            nq = qb.shape[0]
            row[:nq], row[nq:2 * nq], row[2 * nq:] = m.float(), den.float(), num.float().reshape(-1)

        def merge_fn(rows, Bq):
            bank = holder["bank"]
            G, CL, C = rows.shape[0], bank.CL, bank.C
            ms = [rows[k, :Bq].double() for k in range(G)]
            dens = [rows[k, Bq:2 * Bq].double() for k in range(G)]
            nums = []
            for k in range(G):
                full = torch.zeros(Bq, C, dtype=torch.float64)
                lo_k = int(bank.class_lo[k]) if bank.class_lo is not None else 0
                win = rows[k, 2 * Bq:2 * Bq + Bq * CL].double().reshape(Bq, CL)
                full[:, lo_k:lo_k + CL] = win[:, :C - lo_k]
                nums.append(full)
            return O.nw_merge_f64(ms, dens, nums).float()

        holder["bank"] = bank = net.precompute_sharded(partial_fn=partial_fn, merge_fn=merge_fn)
        # the whole bank, the way precompute() orders it, computed locally for the check
        feats, ys = [], []
        for img, label, _ in net.support_eval.support_loaders[0]:
            feats.append(net.featurizer(img).detach())
            ys.append(label)
        full_feat, full_y = torch.cat(feats), torch.cat(ys)
        lo, hi = shard_bounds(len(full_y), world, rank)
        assert torch.equal(bank.y, full_y[lo:hi]) and torch.allclose(bank.feat, full_feat[lo:hi])
        x = torch.randn(5, 3, 4, 4, generator=torch.Generator().manual_seed(3))
        out = net.predict(x, "full")
        ref = O.nw_head_f64(net.featurizer(x).detach(), full_feat, full_y, 10).float()
        q.put((rank, len(full_y), (out - ref).abs().max().item()))
    finally:
        dist.destroy_process_group()


def test_precompute_sharded_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_precompute, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n, err in res:
        assert n == 70 and err < 2e-5, (rank, n, err)
