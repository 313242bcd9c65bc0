"""Alias of nwhead_amd.util under the reference's import path (`from util.metric import support_influence`)."""
import sys
from nwhead_amd.util import metric, utils  # noqa: F401
sys.modules[__name__ + ".metric"] = metric
sys.modules[__name__ + ".utils"] = utils
