"""Cost of an UNSORTED label vector on the cached 'full' path (one run per support row)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nwhead_amd import ops
import bench
B, N, d, C = (int(a) for a in sys.argv[1:5])
dev = torch.device("cuda:0")
q, s, sy = bench.make_inputs(B, N, d, C, dev)
cache = ops.SplitBank(s)
perm = torch.randperm(N, device=dev)
for name, lab in (("class-sorted", sy), ("shuffled", sy[perm])):
    t = bench.time_kernel_events(lambda: ops.nw_head(q, s, lab, C, support_cache=cache), 10, warmup=3)
    print(f"{name:14s} ({B},{N},{d},{C}) fwd {t*1e6:9.1f} us")
