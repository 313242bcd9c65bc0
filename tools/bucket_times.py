"""Per-bucket device time inside one long predict_stream run (K3 shape): is the pace constant?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nwhead_amd import ops
from nwhead_amd.sharded import ShardedBank
dev = torch.device("cuda:0")
B, N, d, C = 256, 50000, 512, 200
q, s, sy = bench.make_inputs(B, N, d, C, dev)
bank = ShardedBank(s, sy, C)
qs = [torch.randn(B, d, device=dev) for _ in range(4)]
qcat = torch.cat([qs[i % 4] for i in range(16)])
for _ in range(2):
    ops.nw_head(qcat, bank.feat, bank.y, C, support_cache=bank.cache)
torch.cuda.synchronize()
time.sleep(0.5)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
t0 = time.perf_counter()
ev[0].record()
for i in range(n):
    ops.nw_head(qcat, bank.feat, bank.y, C, support_cache=bank.cache)
    ev[i + 1].record()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
ts = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]
print("host enqueue %.1f us per launch; wall %.1f us per launch" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
print("first 60:", " ".join("%.0f" % t for t in ts[:60]))
print("last 12: ", " ".join("%.0f" % t for t in ts[-12:]))
