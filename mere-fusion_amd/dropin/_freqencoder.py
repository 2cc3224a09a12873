"""`import _freqencoder` as the reference wrappers do, resolved to the MI355X implementation."""
from mere_fusion_amd.ernerf._freqencoder import *  # noqa: F401,F403
