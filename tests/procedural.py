"""Procedural (closed-form) weights so that backbone fixtures store only inputs and outputs:
every parameter / buffer is a deterministic function of its state_dict NAME and element index."""
import zlib

import torch


def fill_procedural(module):
    sd = module.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            if t.dtype not in (torch.float32, torch.float64):
                continue                      # num_batches_tracked
            k = (zlib.crc32(name.encode()) % 1000) * 0.001
            idx = torch.arange(t.numel(), dtype=torch.float64)
            wave = torch.sin(idx * 0.37 + 6.283 * k)
            if name.endswith("running_var"):
                v = 1.0 + 0.2 * wave * wave
            elif name.endswith("running_mean"):
                v = 0.05 * wave
            elif t.dim() == 1 and name.endswith("weight"):      # BN scale
                v = 1.0 + 0.1 * wave
            elif t.dim() == 1:                                   # BN shift / bias
                v = 0.05 * wave
            else:                                                # conv / linear weight
                fan_in = t[0].numel()
                v = wave * (1.5 / fan_in) ** 0.5
            t.copy_(v.reshape(t.shape).to(t.dtype))
    return module
