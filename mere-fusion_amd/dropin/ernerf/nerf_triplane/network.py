"""app.py:17 `from ernerf.nerf_triplane.network import NeRFNetwork`: the reference's own class with the MI355X render path mixed in in front
(mere_fusion_amd/ernerf/network.py).  Everything else this module exports (`AudioNet`, `AudioAttNet`, `MLP`) is the reference's, untouched."""
from mere_fusion_amd.ernerf.network import HipRenderMixin, load_reference_module

_ref = load_reference_module(__name__, __file__, __package__)
AudioAttNet, AudioNet, MLP = _ref.AudioAttNet, _ref.AudioNet, _ref.MLP


class NeRFNetwork(HipRenderMixin, _ref.NeRFNetwork):
    pass
