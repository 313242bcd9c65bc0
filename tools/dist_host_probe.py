"""Host time per bucket of ShardedBank.predict_stream on the multi-rank code path (one rank, NCCL all-gather of one):
issue time against device time, at the per-rank shape of 8 shards."""
import os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nwhead_amd.sharded import ShardedBank
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
bucket, N, C, B, d = 26, 6250, 25, 256, 512
q, s, sy = bench.make_inputs(B, N, d, C, dev)
bank = ShardedBank(s, sy, C)
bank.world = 1
qbuf = torch.randn(bucket * B, d, device=dev)
qs = [qbuf[k * B:(k + 1) * B] for k in range(bucket)]
import gc; gc.collect(); gc.freeze()
# force the partial + all-gather + merge path
orig = bank._hip_partial
steps = 40 * bucket
def run():
    return bank.predict_stream([qs[i % bucket] for i in range(steps)], bucket=bucket)[-1]
bank._partial = lambda p, q_: orig(p, q_)      # not the identity object: takes the exchange path even with one rank
for rep in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"run {rep}: host issue {1e6 * (t1 - t0) / 40:.1f} us per bucket, total {1e6 * (t2 - t0) / 40:.1f} us per bucket", flush=True)
dist.destroy_process_group()
