"""Datasets for the training harness that need neither torchvision nor a network.

The reference's train.py builds torchvision datasets (train.py:163-190); what NWNet needs from a
dataset is only: `__len__`, `__getitem__ -> (image CHW float tensor, int label)`, `.targets` (list of
labels, nwhead/nw.py:71-72) and `.num_classes` (train.py:200).

* SyntheticImages : procedural class-dependent images (a coloured oriented grating per class plus
                    per-sample noise), deterministic in (seed, index): learnable, tiny, no files.
* CIFARFromDisk   : the python-pickle CIFAR-10/100 batches read straight from `root`
                    (cifar-10-batches-py/ or cifar-100-python/), with the reference's train-time
                    augmentation (random crop 32 pad 4, horizontal flip) and normalisation
                    (train.py:146-157) written in torch.
"""
import math
import os
import pickle

import numpy as np
import torch
from torch.utils.data import Dataset

CIFAR_MEAN = (0.4914, 0.4822, 0.4465)
CIFAR_STD = (0.2023, 0.1994, 0.2010)


class SyntheticImages(Dataset):
    def __init__(self, n_per_class=20, num_classes=10, size=32, seed=0, noise=0.35):
        self.num_classes, self.size, self.seed, self.noise = num_classes, size, seed, noise
        self.targets = [c for c in range(num_classes) for _ in range(n_per_class)]
        yy, xx = torch.meshgrid(torch.linspace(-1, 1, size), torch.linspace(-1, 1, size), indexing="ij")
        protos = []
        for c in range(num_classes):
            ang = math.pi * c / num_classes
            freq = 2.0 + (c % 3)
            wave = torch.sin(freq * math.pi * (xx * math.cos(ang) + yy * math.sin(ang)))
            tint = torch.tensor([math.cos(2.1 * c), math.cos(2.1 * c + 2.1), math.cos(2.1 * c + 4.2)])
            protos.append(wave[None] * (0.6 + 0.4 * tint[:, None, None]))
        self.protos = torch.stack(protos)                      # (C,3,H,W)

    def __len__(self):
        return len(self.targets)

    def __getitem__(self, i):
        i = int(i)
        y = self.targets[i]
        g = torch.Generator().manual_seed(self.seed * 1_000_003 + i)
        return self.protos[y] + self.noise * torch.randn(3, self.size, self.size, generator=g), y


class CIFARFromDisk(Dataset):
    def __init__(self, root, train=True, cifar100=False, augment=None):
        self.train = train
        self.augment = train if augment is None else augment
        self.num_classes = 100 if cifar100 else 10
        if cifar100:
            files = [os.path.join(root, "cifar-100-python", "train" if train else "test")]
            key = b"fine_labels"
        else:
            base = os.path.join(root, "cifar-10-batches-py")
            files = [os.path.join(base, "data_batch_%d" % i) for i in range(1, 6)] if train else \
                    [os.path.join(base, "test_batch")]
            key = b"labels"
        data, labels = [], []
        for f in files:
            if not os.path.exists(f):
                raise FileNotFoundError(f"{f}: CIFAR python batches expected under {root} (no download here)")
            with open(f, "rb") as fh:
                d = pickle.load(fh, encoding="bytes")
            data.append(np.asarray(d[b"data"], dtype=np.uint8).reshape(-1, 3, 32, 32))
            labels += list(d[key])
        self.data = torch.from_numpy(np.concatenate(data))     # (n,3,32,32) uint8
        self.targets = [int(v) for v in labels]
        self.mean = torch.tensor(CIFAR_MEAN).view(3, 1, 1)
        self.std = torch.tensor(CIFAR_STD).view(3, 1, 1)

    def __len__(self):
        return len(self.targets)

    def __getitem__(self, i):
        i = int(i)
        img = self.data[i].float() / 255.0
        if self.augment:
            pad = torch.nn.functional.pad(img, (4, 4, 4, 4))
            dx, dy = np.random.randint(0, 9, size=2)
            img = pad[:, dy:dy + 32, dx:dx + 32]
            if np.random.rand() < 0.5:
                img = img.flip(-1)
        return (img - self.mean) / self.std, self.targets[i]


def build_datasets(name, data_dir, size=32, n_per_class=20, num_classes=10, seed=0, noise=0.35):
    """(train, val) for `--dataset`: synthetic | cifar10 | cifar100."""
    if name == "synthetic":
        return (SyntheticImages(n_per_class, num_classes, size, seed, noise),
                SyntheticImages(max(n_per_class // 2, 2), num_classes, size, seed + 1, noise))
    if name in ("cifar10", "cifar100"):
        c100 = name == "cifar100"
        return CIFARFromDisk(data_dir, True, c100), CIFARFromDisk(data_dir, False, c100)
    raise NotImplementedError(name)
