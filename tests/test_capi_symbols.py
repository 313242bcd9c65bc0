"""The C-ABI library loads on a CPU-only host and exports every symbol include/nwhead_hip.h declares
(no compute calls here: there is no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "nwhead_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(nw_[a-z0-9_]+)\s*\(", hdr)))


def test_header_declares_expected_surface():
    names = _declared()
    for n in ("nw_fwd_f32", "nw_bwd_f32", "nw_scores_f32", "nw_fwd_partial_f32", "nw_merge_finalize_f32",
              "nw_support_influence_f32", "nw_row_norm2_f32", "nw_fwd_workspace_bytes", "nw_bwd_workspace_bytes"):
        assert n in names


def test_library_exports_every_declared_symbol():
    from nwhead_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in nwhead_hip.h but not exported"
    # and the ctypes signature table covers exactly the declared functions
    assert sorted(_lib.SIGNATURES) == _declared()


def test_status_strings_and_argument_checks_without_gpu():
    from nwhead_amd import _lib
    lib = _lib.load()
    assert lib.nw_abi_version() == 2
    assert lib.nw_status_string(0) == b"ok"
    assert b"workspace" in lib.nw_status_string(-3)
    # argument validation happens before any HIP call
    assert lib.nw_scores_f32(None, None, None, -1, 1, 1, 0, None, 0, None) == -1      # negative size
    assert lib.nw_scores_f32(None, None, None, 1, 1, 1, 9, None, 0, None) == -2       # unknown kind
    assert lib.nw_scores_f32(None, None, None, 1, 1, 1, 0, None, 0, None) == -1       # null pointers
    assert lib.nw_fwd_workspace_bytes(256, 10000, 512, 200) > 0
    assert lib.nw_fwd_workspace_bytes(0, 10000, 512, 200) == 0


def test_missing_library_fails_loudly(monkeypatch):
    from nwhead_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libnwhead_hip.so")
    with pytest.raises(_lib.NWHipError):
        _lib.load()


def test_product_path_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "nwhead_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("the oracle", ""), f"{f} references oracle/"


def test_argument_validation_under_address_and_ub_sanitizers():
    """the C ABI's argument validation, built host-only from the library's own sources with
    -fsanitize=address,undefined (`make -C nwhead_amd/csrc sanitize`; no device code, a HIP runtime stand-in that
    answers "no device") and run: every malformed call is refused with its documented status, no sanitizer report."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "nwhead_amd", "csrc"), "sanitize", "-j4"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "all argument checks refused as documented" in r.stdout
    assert "runtime error" not in r.stdout + r.stderr and "AddressSanitizer" not in r.stdout + r.stderr
