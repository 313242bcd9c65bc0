"""Forward time per score kind at one shape (cached split-fp16 bank)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nwhead_amd import ops
import bench
B, N, d, C = (int(a) for a in sys.argv[1:5])
dev = torch.device("cuda:0")
q, s, sy = bench.make_inputs(B, N, d, C, dev)
cache = ops.SplitBank(s)
ls = torch.tensor(float(np.log(1 / 0.07)), device=dev)
for kind in ("euclidean", "hypersphere_euclidean", "cosine", "dotproduct", "clip"):
    t = bench.time_kernel_events(lambda: ops.nw_head(q, s, sy, C, kind, ls if kind == "clip" else None, support_cache=cache), 30, warmup=5)
    print(f"{kind:24s} ({B},{N},{d},{C}) fwd {t*1e6:8.1f} us  {2*B*N*d/t/1e12:6.1f} TFLOP/s")
