"""`import _gridencoder` as the reference wrappers do, resolved to the MI355X implementation."""
from mere_fusion_amd.ernerf._gridencoder import *  # noqa: F401,F403
