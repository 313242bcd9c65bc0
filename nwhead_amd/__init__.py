"""nwhead_amd -- MI355X-native Nadaraya-Watson head (drop-in for alanqrwang/nwhead's hot path).

Layout:
    csrc/          hand-written HIP kernels + the C ABI (include/nwhead_hip.h)
    _lib.py        ctypes binding (no fallback: raises when the .so is missing)
    ops.py         tensor-level ops + autograd node
    nwhead/        mirror of the reference's ``nwhead`` package API (NWNet, NWHead, get_kernel, ...)
    model/         mirror of ``model.load_model`` and the backbones that feed the head
    util/          mirror of ``util.metric.support_influence``
    sharded.py     support bank sharded over ranks, RCCL merge of (m, den, num) partials
"""
from ._lib import LIB_PATH, NWHipError, SCORE_KINDS  # noqa: F401

__all__ = ["LIB_PATH", "NWHipError", "SCORE_KINDS"]
