"""The sharded 'full' inference path (nwhead_amd/sharded.py) beyond two ranks, on CPU (gloo; compute hooks = the oracle):
world 8 with a K3-like class-sorted bank scaled down, world 3 with N % G != 0, world 8 with EMPTY shards (N < G), and a
bank whose classes are so unequal that one rank's class window is much wider than the others' -- all through
ShardedBank.__init__ (the all-gather of the (lo, hi) class boxes) and predict_stream."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _labels(case, N, C):
    if case == "balanced":                       # K3's shape: C classes of N / C rows, class-sorted
        return (torch.arange(N) * C // N)
    if case == "skewed":                         # one huge class, then many tiny ones: the last ranks span many classes
        big = N // 2
        rest = torch.arange(N - big) * (C - 1) // max(N - big, 1) + 1
        return torch.cat((torch.zeros(big, dtype=torch.int64), rest))
    raise ValueError(case)


def _worker(rank, world, port, q, case, N, C):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from nwhead_amd.sharded import ShardedBank, shard_bounds
        from oracle import nw_oracle as O
        torch.set_num_threads(1)
        g = torch.Generator().manual_seed(17)
        d, B = 16, 5
        s = torch.randn(N, d, generator=g)
        sy = _labels(case, N, C)
        batches = [torch.randn(B, d, generator=g) for _ in range(3)] + [torch.randn(2, d, generator=g)]
        lo, hi = shard_bounds(N, world, rank)
        holder = {}

        def partial_fn(row, qb):
            bank = holder["bank"]
            nq = qb.shape[0]
            if hi == lo:                                   # an empty shard: m = -inf, den = 0, num = 0
                row[:nq], row[nq:2 * nq], row[2 * nq:] = float("-inf"), 0.0, 0.0
                return
            m, den, num = O.nw_partials_f64(qb, s[lo:hi], bank.y_local, bank.CL)
            row[:nq], row[nq:2 * nq], row[2 * nq:] = m.float(), den.float(), num.float().reshape(-1)

        def merge_fn(rows, Bq):
            bank = holder["bank"]
            G, CL = rows.shape[0], bank.CL
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

        bank = holder["bank"] = ShardedBank(s[lo:hi], sy[lo:hi], C, partial_fn=partial_fn, merge_fn=merge_fn)
        outs = bank.predict_stream(batches, bucket=2)
        ref = [O.nw_head_f64(qb, s, sy, C).float() for qb in batches]
        err = max((o - r).abs().max().item() for o, r in zip(outs, ref))
        # every rank agrees on the window width; windows cover the rank's classes
        cl = torch.tensor([bank.CL], dtype=torch.int64)
        allcl = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allcl, cl)
        assert len({int(c) for c in allcl}) == 1
        if hi > lo and bank.class_lo is not None:
            assert int(bank.class_lo[rank]) <= int(sy[lo]) and int(sy[hi - 1]) < int(bank.class_lo[rank]) + bank.CL
        q.put((rank, bank.CL, hi - lo, err))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case,N,C", [
    (8, "balanced", 403, 40),       # K3 scaled down: 8 shards of 50-51 rows, ~5 classes each (class windows of 6-7)
    (3, "balanced", 100, 7),        # N % G != 0
    (8, "balanced", 5, 4),          # N < G: three ranks hold nothing
    (4, "skewed", 240, 61),         # rank 0-1: one class; ranks 2-3: 30 classes each -> one wide window for all
])
def test_sharded_predict_many_ranks(world, case, N, C):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, case, N, C)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len({cl for _, cl, _, _ in res}) == 1
    assert sum(n for _, _, n, _ in res) == N
    for rank, cl, n, err in res:
        assert err < 2e-5, (rank, cl, n, err)
    if case == "skewed":
        assert res[0][1] >= 30          # the widest window (a tiny-class rank) sets CL for every rank
    if N < world:
        assert sum(1 for _, _, n, _ in res if n == 0) == world - N
