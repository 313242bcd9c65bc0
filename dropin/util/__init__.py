"""Alias of nwhead_amd.util under the reference's import path (`from util.metric import support_influence`)."""
import sys
from nwhead_amd.util import metric  # noqa: F401
sys.modules[__name__ + ".metric"] = metric
