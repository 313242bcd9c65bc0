"""The multi-rank product path with the real kernels: two ranks (gloo, both on cuda:0 -- RCCL refuses two
ranks on one device, and the test box has one) shard a class-sorted bank, run nw_fwd_partial_f32 on their
shards with class windows, all-gather the packed partials and merge them with nw_merge_finalize_f32."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NW_SPLIT_ALWAYS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from nwhead_amd.sharded import ShardedBank, shard_bounds
        from oracle import nw_oracle as O
        dev = torch.device("cuda:0")
        g = torch.Generator().manual_seed(17)
        B, N, d, C = 96, 4001, 64, 9
        s = torch.randn(N, d, generator=g)
        sy = (torch.arange(N) % C).sort().values
        batches = [torch.randn(B, d, generator=g) for _ in range(5)]
        lo, hi = shard_bounds(N, world, rank)
        bank = ShardedBank(s[lo:hi].to(dev), sy[lo:hi].to(dev), C)
        assert bank.class_lo is not None and bank.CL < C          # class windows are in use
        outs = bank.predict_stream([b.to(dev) for b in batches], bucket=2)
        ref = [O.nw_head_f64(b, s, sy, C).float() for b in batches]
        err = max((o.cpu() - r).abs().max().item() for o, r in zip(outs, ref))
        q.put((rank, len(outs), err))
    finally:
        dist.destroy_process_group()


def test_sharded_hip_two_ranks_one_device():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, n, err in res:
        assert n == 5 and err < 5e-5, (rank, n, err)
