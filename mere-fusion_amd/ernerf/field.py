"""The ER-NeRF radiance field on MI355X: `NeRFNetwork.forward` of ernerf/nerf_triplane/network.py:249-277 (+ `density`
:280-308, `encode_x` :211-219) behind the same call signature, backed by mf_nerf_field_* of libmerefusion_hip.so.

    field = HipNeRFField(model.state_dict(), bound=opt.bound, individual_dim=opt.ind_dim, exp_eye=opt.exp_eye)
    model.forward = field.forward          # the render loop calls self.forward(xyzs, dirs, enc_a, ind_code, eye), renderer.py:260
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib

LN2 = float(np.log(2.0))


def grid_geometry(num_levels=12, base_resolution=64, log2_hashmap_size=14, desired_resolution=512, input_dim=2, align_corners=False):
    """offsets and per_level_scale exactly as GridEncoder.__init__ computes them (grid.py:94-123)."""
    per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    offsets, offset = [], 0
    max_params = 2 ** log2_hashmap_size
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        n = min(max_params, (resolution if align_corners else resolution + 1) ** input_dim)
        n = int(np.ceil(n / 8) * 8)
        offsets.append(offset)
        offset += n
    offsets.append(offset)
    return np.array(offsets, np.int32), float(per_level_scale)


class HipNeRFField:
    def __init__(self, state_dict, bound=1.0, individual_dim=4, exp_eye=True, precision="bf16x3", max_samples=512 * 512, device="cuda"):
        self.device = torch.device(device)
        _lib.init_device(self.device.index or 0)
        self._lib = _lib.lib()
        offsets, pls = grid_geometry(desired_resolution=512 * bound)
        cfg = _lib.MfNerfFieldConfig(bound=float(bound), num_levels=12, level_dim=1, base_resolution=64, log2_per_level_scale=float(np.log2(pls)),
                                     audio_dim=32, geo_feat_dim=64, hidden_dim=64, individual_dim=int(individual_dim), exp_eye=int(bool(exp_eye)))
        for i, o in enumerate(offsets):
            cfg.offsets[i] = int(o)
        keep = {k: v for k, v in state_dict.items() if k.split(".")[0] in ("encoder_xy", "encoder_yz", "encoder_xz", "sigma_net", "color_net",
                                                                              "aud_ch_att_net", "eye_att_net") and k.endswith(("embeddings", "weight"))}
        arr, self._keep = _lib.tensor_array(keep)
        self._h = C.c_void_p()
        _lib.check(self._lib.mf_nerf_field_create(C.byref(cfg), arr, len(arr), _lib.PRECISIONS[precision], int(max_samples), C.byref(self._h)),
                   "mf_nerf_field_create")
        self.exp_eye, self.individual_dim, self.max_samples = bool(exp_eye), int(individual_dim), int(max_samples)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.mf_nerf_field_destroy(h)
            self._h = None

    def forward(self, x, d, enc_a, c, e=None):
        """x, d: [M, 3]; enc_a: [1, 32]; c: [1, ind_dim] or None; e: [1, 1] eye feature.  Returns the reference's tuple
        (sigma [M], color [M, 3], ambient_aud [M, 1], ambient_eye [M, 1], uncertainty [M, 1]); the reference's uncertainty is a
        [M, 36, 1] tensor of ln 2 of which the compositor reads the first M floats (network.py:240-246) -- same values here."""
        for t, name in ((x, "x"), (d, "d"), (enc_a, "enc_a")):
            if not (torch.is_tensor(t) and t.is_cuda):
                raise RuntimeError(f"HipNeRFField.forward: {name} must be a CUDA tensor (there is no CPU path)")
        if self.exp_eye and e is None:
            raise RuntimeError("HipNeRFField.forward: the field was built with exp_eye; pass the eye feature")
        M = x.shape[0]
        x = x.float().contiguous(); d = d.float().contiguous()
        ea = enc_a.float().reshape(-1).contiguous()
        cc = c.float().reshape(-1).contiguous() if (c is not None and self.individual_dim) else None
        sig = torch.empty(M, device=x.device)
        rgb = torch.empty(M, 3, device=x.device)
        aa, ae, un = (torch.empty(M, 1, device=x.device) for _ in range(3))
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        eye = float(e.reshape(-1)[0]) if e is not None else 0.0
        _lib.check(self._lib.mf_nerf_field_forward(self._h, p(x), p(d), p(ea), p(cc), eye, M, p(sig), p(rgb), p(aa), p(ae), p(un),
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)), "mf_nerf_field_forward")
        return sig, rgb, aa, ae, un

    __call__ = forward
