"""`musetalk.utils.blending` (musetalk/utils/blending.py): `get_image_blending` -- the per-frame paste-back of
musereal.py:238-247 -- runs on the GPU; the offline avatar-preparation helpers (`get_image`, `get_image_prepare_material`, which need
the BiSeNet face parser) are forwarded to the reference's module on first use."""
from mere_fusion_amd.musetalk.utils.blending import get_image_blending, get_crop_box  # noqa: F401


import functools


@functools.lru_cache(maxsize=1)
def _reference_module():
    """Loaded ONCE: the reference's module builds `FaceParsing()` (the BiSeNet weights) at import time (blending.py:8), and these helpers are called per
    avatar frame."""
    import importlib.util
    import os
    import sys
    for d in sys.path:
        cand = os.path.join(d, "musetalk", "utils", "blending.py")
        if os.path.isfile(cand) and os.path.abspath(cand) != os.path.abspath(__file__):
            spec = importlib.util.spec_from_file_location("musetalk.utils._reference_blending", cand)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            return mod
    raise ImportError("the reference's musetalk/utils/blending.py is not on sys.path (needed only for avatar preparation)")


def get_image(*a, **k):
    return _reference_module().get_image(*a, **k)


def get_image_prepare_material(*a, **k):
    return _reference_module().get_image_prepare_material(*a, **k)
