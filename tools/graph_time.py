"""Eager vs hipGraph replay of one head forward at small shapes (host-bound territory)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nwhead_amd import ops
import bench
dev = torch.device("cuda:0")
for B, N, d, C in [(64, 1000, 512, 200), (256, 10000, 512, 200), (256, 50000, 512, 200)]:
    q, s, sy = bench.make_inputs(B, N, d, C, dev)
    bank = ops.SplitBank(s, sy)
    fn = lambda: ops.nw_head(q, s, sy, C, support_cache=bank)
    ref = fn()
    te = bench.time_kernel_events(fn, 200, warmup=20)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    g.replay(); torch.cuda.synchronize()
    assert torch.equal(out, ref), (out - ref).abs().max()
    tg = bench.time_kernel_events(g.replay, 200, warmup=20)
    print(f"({B},{N},{d},{C}) eager {te*1e6:7.2f} us   graph replay {tg*1e6:7.2f} us", flush=True)
