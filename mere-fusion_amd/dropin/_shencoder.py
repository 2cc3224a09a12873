"""`import _shencoder` as the reference wrappers do, resolved to the MI355X implementation."""
from mere_fusion_amd.ernerf._shencoder import *  # noqa: F401,F403
