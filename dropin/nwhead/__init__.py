"""Alias of nwhead_amd.nwhead under the reference's import path (`from nwhead.nw import NWNet`)."""
import sys
from nwhead_amd.nwhead import kernel, nw, support, utils  # noqa: F401
from nwhead_amd.nwhead import NWHead, NWNet, get_kernel  # noqa: F401
for _m in ("kernel", "nw", "support", "utils"):
    sys.modules[__name__ + "." + _m] = getattr(sys.modules["nwhead_amd.nwhead"], _m)
