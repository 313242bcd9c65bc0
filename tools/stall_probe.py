"""Per-call wall time of ShardedBank.predict_stream at a bench shape: looks for the 30-40 ms stalls seen in the timed region."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nwhead_amd.sharded import ShardedBank
bucket, N, C = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B, d = 256, 512
dev = torch.device("cuda:0")
q, s, sy = bench.make_inputs(B, N, d, C, dev)
bank = ShardedBank(s, sy, C)
qbuf = torch.randn(bucket * B, d, device=dev)
qs = [qbuf[k * B:(k + 1) * B] for k in range(bucket)]
def run(nsteps):
    return bank.predict_stream([qs[i % bucket] for i in range(nsteps)], bucket=bucket)[-1]
steps = -(-832 // bucket) * bucket
import gc
if os.environ.get('NOGC') == '1':
    gc.collect(); gc.disable()
for rep in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    run(steps)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"run {rep}: issue {1e3 * (t1 - t0):7.2f} ms, total {1e3 * (t2 - t0):7.2f} ms  ({steps} steps, {steps // bucket} launches)", flush=True)
stats = torch.cuda.memory_stats()
print("allocator: num_alloc_retries", stats.get("num_alloc_retries"), "segments", stats.get("segment.all.current"), "reserved MB", stats.get("reserved_bytes.all.current", 0) >> 20,
      "device mallocs", stats.get("num_device_alloc"), "device frees", stats.get("num_device_free"))
