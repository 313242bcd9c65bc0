"""INTEGRATION.md section B, executed: the ctypes stub a maintainer of the reference would add binds
libnwhead_hip.so exactly as printed there and reproduces NWHead.forward."""
import os
import re

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_stub_runs_as_printed():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from nwhead_amd import _lib, ops
    from oracle import nw_oracle as O
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(# nwhead/_hip\.py.*?)```", text, re.S).group(1)
    code = code.replace('C.CDLL("libnwhead_hip.so")', f'C.CDLL({_lib.LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)

    class EuclideanDistance:                       # the reference's kernel class name (nwhead/kernel.py:13)
        pass

    class CosineDistance:
        pass

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    B, N, d, C = 40, 900, 64, 7
    x, sx = torch.randn(B, d, generator=g), torch.randn(N, d, generator=g)
    sy = (torch.arange(N) % C).sort().values
    for kern, kind in ((EuclideanDistance(), "euclidean"), (CosineDistance(), "cosine")):
        ref = O.nw_head_f64(x, sx, sy, C, kind).numpy()
        plain = ns["nw_forward"](kern, C, x.to(dev), sx.to(dev), sy.to(dev))
        bank = ns["prepare_bank"](sx.to(dev))
        fast = ns["nw_forward"](kern, C, x.to(dev), sx.to(dev), sy.to(dev), bank=bank)
        for out in (plain, fast):
            assert out.shape == (B, C)
            assert abs(out.cpu().double().numpy() - ref).max() < 5e-5


def test_build_then_smoke_in_one_process():
    """The library may be loaded before anything has imported torch (build() does): _lib.load() must bring
    torch's own HIP runtime in first, or the process gets two runtimes and the device check fails."""
    import subprocess
    import sys
    code = ("import __graft_entry__ as g; g.build(); from nwhead_amd import _lib; import torch; "
            "assert torch.cuda.is_available(); _lib.check(_lib.load().nw_device_check(), 'nw_device_check'); print('ok')")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
