"""One rank of tests/test_guards_gpu.py::test_rccl_two_ranks_sharded_predict (launched by torch.distributed.run):
backend "nccl" (RCCL), one device per rank, ShardedBank.predict_stream against the fp64 oracle."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
    dev = torch.device("cuda", torch.cuda.current_device())
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from nwhead_amd.sharded import ShardedBank, shard_bounds
        from oracle import nw_oracle as O
        g = torch.Generator().manual_seed(5)
        B, N, d, C = 96, 9000, 128, 40
        s = torch.randn(N, d, generator=g)
        sy = (torch.arange(N) % C).sort().values
        batches = [torch.randn(B, d, generator=g) for _ in range(5)] + [torch.randn(B - 31, d, generator=g)]
        lo, hi = shard_bounds(N, world, rank)
        bank = ShardedBank(s[lo:hi].to(dev), sy[lo:hi].to(dev), C)
        outs = bank.predict_stream([qb.to(dev) for qb in batches], bucket=2)
        torch.cuda.synchronize()
        for qb, o in zip(batches, outs):
            ref = O.nw_head_f64(qb[:16], s, sy, C)
            err = (o[:16].cpu().double() - ref).abs().max().item()
            assert o.shape == (len(qb), C) and err < 3e-5, err
        gathered = [torch.empty_like(outs[0]) for _ in range(world)]
        dist.all_gather(gathered, outs[0])
        assert all(torch.equal(gathered[0], t) for t in gathered)      # every rank merges to the same bits
        print("rccl-ok", rank, flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
