"""app.py:16 `from ernerf.nerf_triplane.utils import *`, provider.py / renderer.py `from .utils import get_rays, ...`: every name of the reference's own module
(loaded from the next directory of the package path, untouched) -- with two of them leaner at inference time (mere_fusion_amd/ernerf/frontend.py):

  get_rays                      the whole-frame case computes what depends on (H, W, intrinsics) once, with the reference's own function; same bits
  Trainer.test_gui_with_data    resize in one launch (`mf_nerf_resize_frame`), pinned device -> host copies with one synchronisation

Every other call -- training, sampled rays, rects -- reaches the reference's code."""
from mere_fusion_amd.ernerf import frontend as _fe
from mere_fusion_amd.ernerf.network import load_reference_module as _load_reference_module

_ref = _load_reference_module(__name__, __file__, __package__)
globals().update({_k: _v for _k, _v in vars(_ref).items() if not (_k.startswith("__") and _k.endswith("__"))})      # `import *` exports what the reference's module exports

_reference_get_rays = _ref.get_rays


def get_rays(poses, intrinsics, H, W, N=-1, patch_size=1, rect=None):
    return _fe.get_rays(_reference_get_rays, poses, intrinsics, H, W, N, patch_size, rect)


class Trainer(_fe.TrainerMixin, _ref.Trainer):
    _mf_linear_to_srgb = staticmethod(_ref.linear_to_srgb)
