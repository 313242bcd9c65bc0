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


def procedural_input(n, c, h, w, key=0):
    """A closed-form (n, c, h, w) fp32 batch for fixtures that store only outputs: every element a function of its flat index
    (integer hash -> [-1, 1), fp64 arithmetic, then rounded to fp32: bit-identical wherever it is regenerated), with a
    per-image offset and a smooth component so that images and channels differ the way data does."""
    idx = torch.arange(n * c * h * w, dtype=torch.int64)
    hsh = (idx * 2654435761 + 40503 * (key + 1)) % 4294967296
    hsh = (hsh ^ (hsh >> 15)) * 2246822519 % 4294967296
    u = hsh.double() / 4294967296.0 * 2.0 - 1.0
    img = (idx // (c * h * w)).double()
    smooth = torch.sin(idx.double() * 0.0137 + img * 0.7)
    return (1.2 * u + 0.5 * smooth + 0.1 * torch.cos(img * 1.3)).reshape(n, c, h, w).float()


def fill_procedural_hash(module):
    """Like fill_procedural, but every weight an integer HASH of (name, index) -> uniform, scaled like a Kaiming
    initialisation: the sine waves of fill_procedural make structured filters whose feature maps hold near-constant channels
    (variance ~1e-6), and training-mode BatchNorm then amplifies fp32 rounding a thousandfold -- unusable for a tight
    train-mode fixture (G6b).  Bit-identical wherever it runs (int64 and fp64 arithmetic only)."""
    sd = module.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            if t.dtype not in (torch.float32, torch.float64):
                continue
            k = zlib.crc32(name.encode())
            idx = torch.arange(t.numel(), dtype=torch.int64)
            hsh = (idx * 2654435761 + k) % 4294967296
            hsh = (hsh ^ (hsh >> 15)) * 2246822519 % 4294967296
            hsh = (hsh ^ (hsh >> 13)) * 3266489917 % 4294967296
            u = (hsh ^ (hsh >> 16)).double() / 4294967296.0 * 2.0 - 1.0          # uniform [-1, 1)
            if name.endswith("running_var"):
                v = 1.0 + 0.2 * u * u
            elif name.endswith("running_mean"):
                v = 0.05 * u
            elif t.dim() == 1 and name.endswith("weight"):      # BN scale
                v = 1.0 + 0.2 * u
            elif t.dim() == 1:                                   # BN shift / bias
                v = 0.1 * u
            else:                                                # conv / linear weight: variance 2 / fan_in
                v = u * (6.0 / t[0].numel()) ** 0.5
            t.copy_(v.reshape(t.shape).to(t.dtype))
    return module
