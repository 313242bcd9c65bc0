import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# The C ABI routes small problems to the fp32 path even when a SplitBank is supplied (capi.hip,
# split_pays); the parity tests want the split-fp16 kernels on their small shapes, so pin the choice
# before the library is loaded (it reads the variable once).
os.environ.setdefault("NW_SPLIT_ALWAYS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    return torch.cuda.is_available()
