"""Mirror of the reference's ``nwhead`` package surface (nwhead/nw.py, kernel.py, support.py, utils.py)."""
from .kernel import get_kernel  # noqa: F401
from .nw import NWHead, NWNet  # noqa: F401
