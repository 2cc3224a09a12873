"""`import _raymarching_face` as the reference wrappers do, resolved to the MI355X implementation."""
from mere_fusion_amd.ernerf._raymarching_face import *  # noqa: F401,F403
