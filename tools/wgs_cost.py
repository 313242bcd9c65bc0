"""What leaving one CU per XCD to a concurrent RCCL kernel costs the persistent tile kernel: the per-rank
workloads of G = 1 / 2 / 4 / 8 shards of the K3 bank (rows and classes divided by G, 26 batches of 256 queries per launch) with
persistent_wgs = 256 and 248, alternated on one device.  usage: python tools/wgs_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nwhead_amd.sharded import ShardedBank
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for G in (1, 2, 4, 8):
    N, C = 50000 // G, 200 // G
    s = torch.randn(N, 512, generator=g).to(dev)
    sy = (torch.arange(N) * C // N).to(dev)
    qs = [torch.randn(256, 512, generator=g).to(dev) for _ in range(26)]
    banks = {w: ShardedBank(s, sy, C, persistent_wgs=w) for w in (0, 248)}
    res = {}
    for rep in range(3):
        for w, bank in banks.items():
            t = bench.time_kernel_events(lambda: bank.predict_stream(qs, bucket=26), 30, warmup=10)
            res.setdefault(w, []).append(t * 1e6)
    a, b = sorted(res[0])[1], sorted(res[248])[1]
    print(f"G={G}: N={N} per rank: 256 workgroups {a:.1f} us per launch, 248 workgroups {b:.1f} us (+{100 * (b / a - 1):.1f} %)", flush=True)
