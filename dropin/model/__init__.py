"""Alias of nwhead_amd.model under the reference's import path (`from model import load_model`)."""
from nwhead_amd.model import load_model  # noqa: F401
