"""K3 launch time and accuracy vs the number of mantissa bits kept in the low fp16 halves
(NW_SPLIT_LBITS is read once per process: run one value per process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nwhead_amd import ops
from oracle import nw_oracle as O
import bench
dev = torch.device("cuda:0")
B, N, d, C = 4096, 50000, 512, 200
q, s, sy = bench.make_inputs(B, N, d, C, dev)
bank = ops.SplitBank(s, sy)
fn = lambda: ops.nw_head(q, s, sy, C, support_cache=bank)
out = fn()
rows = torch.arange(0, B, B // 8)
ref = O.nw_head_f64(q[rows].cpu(), s.cpu(), sy.cpu(), C)
err = (out[rows].cpu().double() - ref).abs().max().item()
t = bench.time_kernel_events(fn, 60, warmup=40, min_warm_ms=60)
print(f"LBITS={os.environ.get('NW_SPLIT_LBITS','10')}: {t*1e6:8.1f} us/launch   max|err| vs fp64 on 8 rows {err:.2e}", flush=True)
