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
sn2 = ops.row_norm2(s)
t = bench.time_kernel_events(lambda: ops.nw_head(q, s, sy, C, support_norm2=sn2), 200, warmup=20)
ts = bench.time_kernel_events(lambda: ops.nw_head(q, s, sy, C), 200, warmup=20)
print(f"RS={os.environ.get('NW_TILE_RS','auto'):>4s} shape=({B},{N},{d},{C}) fwd={t*1e6:8.2f} us  {2*B*N*d/t/1e12:6.1f} TF ({2*B*N*d/t/1e12/157.3*100:4.1f}%)   no-cached-norms={ts*1e6:8.2f} us {2*B*N*d/ts/1e12:6.1f} TF")
