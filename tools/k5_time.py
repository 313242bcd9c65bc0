"""K5 (support_influence) at BASELINE's shape and at B = 4096, and the fused forward + influence: bench.py's own legs, alone.
usage (GPU box, repo root): python tools/k5_time.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
r = bench.measure_influence(256, 10000, 200, dev)
print("K5 256x10000:", json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if "note" not in k and "rotation" not in k}))
r = bench.measure_influence(4096, 10000, 200, dev, iters=20)
print("K5 4096x10000:", json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if "note" not in k and "rotation" not in k}))
r = bench.measure_forward_influence(256, 10000, 512, 200, dev)
print("fused:", json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}))
