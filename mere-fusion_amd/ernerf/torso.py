"""The ER-NeRF torso branch on MI355X: `NeRFRenderer.run_torso` (ernerf/nerf_triplane/renderer.py:294-352) with
`NeRFNetwork.forward_torso` (network.py:166-201) underneath, backed by mf_nerf_torso_* of libmerefusion_hip.so.

    torso = HipTorso(model.state_dict(), torso_shrink=opt.torso_shrink, individual_dim=opt.ind_dim_torso)
    bg = torso.run_torso(bg_coords, poses, bg_color)["bg_color"]         # what run_cuda mixes the head over, renderer.py:272-275

Every pixel runs through the two small MLPs and the occupancy mask is applied in the final mix (no boolean-mask gather/scatter, no
host sync); the frequency-encoded wrapped anchors and the individual code are per-frame constants folded into first-layer biases."""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from .field import grid_geometry


def freq_encode_host(x, degree):
    """FreqEncoder.forward (freq.py:66-78 -> kernel_freq, freqencoder.cu:30-58) for a handful of values, in float32 like the kernel: output c >= D is
    sin(x[c % D] * 2^(col // 2) + (col % 2) * pi / 2) with col = c // D - 1.  (Vectorised: this runs on the host once per frame.)"""
    x = np.asarray(x, np.float32).reshape(-1)
    D = x.shape[0]
    c = np.arange(D, D + 2 * D * degree)
    col, d = c // D - 1, c % D
    arg = np.ldexp(x[d], col // 2).astype(np.float32) + ((col % 2).astype(np.float32) * np.float32(np.float32(np.pi) / 2)).astype(np.float32)
    return np.concatenate([x, np.sin(arg.astype(np.float32), dtype=np.float32)])


def wrapped_anchor_code(anchor_points, poses, ind_code):
    """network.py:175-178: anchors through the inverse pose, perspective-divided, FreqEncoder(6, 3), then the individual code."""
    a = anchor_points.numpy() if torch.is_tensor(anchor_points) else np.asarray(anchor_points, np.float32)
    p = (poses.detach().float().cpu().numpy() if torch.is_tensor(poses) else np.asarray(poses, np.float32)).reshape(4, 4)
    w = torch.from_numpy(a)[None, ...] @ torch.from_numpy(p)[None].permute(0, 2, 1).inverse()          # (torch's own 4 x 4 inverse: the reference's arithmetic)
    w = (w[:, :, :2] / w[:, :, 3, None] / w[:, :, 2, None]).reshape(-1)
    enc = freq_encode_host(w.numpy(), 3)
    code = np.zeros(0, np.float32) if ind_code is None else (ind_code.numpy() if torch.is_tensor(ind_code) and not ind_code.is_cuda else torch.as_tensor(ind_code).detach().float().cpu().numpy()).reshape(-1)
    return np.concatenate([enc, code]).astype(np.float32)


class HipTorso:
    def __init__(self, state_dict, torso_shrink=0.8, individual_dim=8, density_thresh_torso=0.01, mean_density_torso=0.0, grid_size=128,
                 precision="bf16x3", max_pixels=512 * 512, device="cuda"):
        self.device = torch.device(device)
        _lib.init_device(self.device.index or 0)
        self._lib = _lib.lib()
        offsets, pls = grid_geometry(num_levels=16, base_resolution=16, log2_hashmap_size=16, desired_resolution=2048)   # network.py:158
        cfg = _lib.MfNerfTorsoConfig(torso_shrink=float(torso_shrink), num_levels=16, level_dim=2, base_resolution=16,
                                     log2_per_level_scale=float(np.log2(pls)), individual_dim=int(individual_dim), grid_size=int(grid_size))
        for i, o in enumerate(offsets):
            cfg.offsets[i] = int(o)
        keep = {k: v for k, v in state_dict.items() if k.startswith(("torso_deform_net.", "torso_net.", "torso_encoder.embeddings", "density_grid_torso"))}
        arr, self._keep = _lib.tensor_array(keep)
        self._h = C.c_void_p()
        _lib.check(self._lib.mf_nerf_torso_create(C.byref(cfg), arr, len(arr), _lib.PRECISIONS[precision], int(max_pixels), C.byref(self._h)),
                   "mf_nerf_torso_create")
        self.anchor_points = state_dict["anchor_points"].detach().float().cpu()
        codes = state_dict.get("individual_codes_torso")
        self.ind_code = codes[0].detach().float().cpu() if (codes is not None and individual_dim > 0) else None      # renderer.py:318-319
        self.thresh = float(min(density_thresh_torso, mean_density_torso))                                           # renderer.py:325

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.mf_nerf_torso_destroy(h)
            self._h = None

    @torch.no_grad()
    def run_torso(self, bg_coords, poses, bg_color=None):
        """bg_coords: [N, 2] (or [1, N, 2]) CUDA fp32 in [-1, 1]; poses: [1, 4, 4]; bg_color: [N, 3] / [3] tensor, scalar or None (= 1)."""
        if not (torch.is_tensor(bg_coords) and bg_coords.is_cuda):
            raise RuntimeError("HipTorso.run_torso: bg_coords must be a CUDA tensor (there is no CPU path)")
        xy = bg_coords.contiguous().view(-1, 2).float()
        N = xy.shape[0]
        consts = wrapped_anchor_code(self.anchor_points, poses, self.ind_code)
        cbuf = (C.c_float * len(consts))(*consts.tolist())
        out = torch.empty(N, 3, device=xy.device)
        alpha = torch.empty(N, device=xy.device)
        deform = torch.empty(N, 2, device=xy.device)
        bg = bg_color.float().contiguous() if torch.is_tensor(bg_color) else None
        per_ray = bg is not None and bg.numel() == 3 * N
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(self._lib.mf_nerf_torso_forward(self._h, p(xy), cbuf, p(bg), int(per_ray), float(1.0 if bg_color is None else (0.0 if bg is not None else bg_color)),
                                                   self.thresh, N, p(out), p(alpha), p(deform), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   "mf_nerf_torso_forward")
        return {"bg_color": out, "torso_alpha": alpha.view(N, 1), "deform": deform}
