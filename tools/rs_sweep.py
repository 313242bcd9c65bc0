#!/usr/bin/env python3
"""Time the forward at one shape; tile height from env NW_TILE_RS (0 = auto)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nwhead_amd import ops
import bench
B, N, d, C = (int(a) for a in sys.argv[1:5])
dev = torch.device("cuda:0")
q, s, sy = bench.make_inputs(B, N, d, C, dev)
cache = ops.SplitBank(s)
t = bench.time_kernel_events(lambda: ops.nw_head(q, s, sy, C, support_cache=cache), 100, warmup=10)
t32 = bench.time_kernel_events(lambda: ops.nw_head(q, s, sy, C, support_norm2=cache.norm2), 100, warmup=10)
ts = bench.time_kernel_events(lambda: ops.nw_head(q, s, sy, C), 100, warmup=10)
print(f"RS={os.environ.get('NW_TILE_RS','auto'):>4s} shape=({B},{N},{d},{C}) fwd={t*1e6:8.2f} us  {2*B*N*d/t/1e12:6.1f} TF ({2*B*N*d/t/1e12/157.3*100:4.1f}%)   fp32-mfma+norms={t32*1e6:8.2f} us  generic={ts*1e6:8.2f} us {2*B*N*d/ts/1e12:6.1f} TF")
