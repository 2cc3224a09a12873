"""The two per-frame pieces of the reference's ER-NeRF loop that sit either side of `model.render`, leaner, with the same results:

`get_rays` (ernerf/nerf_triplane/utils.py:255-341, called by `NeRFDataset_Test.collate` provider.py:302 once per frame).  The reference rebuilds the pixel
grid, the normalised camera-space directions and the index tensors from scratch on every call -- about twenty small torch launches for values that depend on
(H, W, intrinsics) only.  Here they are computed ONCE per (H, W, intrinsics, device) BY THE REFERENCE'S OWN FUNCTION (called with an identity pose: a product
with the identity is exact, so its `rays_d` are the directions, bit for bit) and a frame costs what depends on the pose: one batched matrix product -- the
reference's own expression `directions @ poses[:, :3, :3].transpose(-1, -2)` -- and a broadcast view.  Same bits as the reference's call.

`Trainer.test_gui_with_data` (utils.py:1190-1223, called by nerfreal.py:110).  Same contract -- {'image': float32 [H, W, 3] numpy, 'depth': [H, W]} -- with the
resize of utils.py:1208-1209 in ONE launch (`mf_nerf_resize_frame`: image bilinear, depth nearest; skipped when the sizes agree, where both are the identity) and
the two device -> host copies into pinned buffers with one synchronisation instead of two pageable `.cpu()` round trips.  The returned arrays are views of a
small ring of pinned buffers: they stay valid for the next three frames (nerfreal.py:111 converts the image to uint8 right away)."""
import ctypes as C

import numpy as np
import torch

from .. import _lib

_RAY_CACHE = {}
_RAY_CACHE_MAX = 8


def _intr_key(intrinsics):
    vals = intrinsics.tolist() if hasattr(intrinsics, "tolist") else list(intrinsics)
    return tuple(float(v) for v in vals)


def get_rays(reference_get_rays, poses, intrinsics, H, W, N=-1, patch_size=1, rect=None):
    """The reference's `get_rays` for the whole-frame case (N <= 0, no rect: what the test-time loader asks for); any other call goes to the reference's function."""
    if N > 0 or rect is not None or not torch.is_tensor(poses) or poses.dim() != 3:
        return reference_get_rays(poses, intrinsics, H, W, N, patch_size, rect)
    key = (int(H), int(W), _intr_key(intrinsics), str(poses.device), poses.dtype)
    hit = _RAY_CACHE.get(key)
    if hit is None:
        eye = torch.eye(4, dtype=poses.dtype, device=poses.device)[None]
        with torch.no_grad():
            r = reference_get_rays(eye, intrinsics, H, W, -1, patch_size, None)
        hit = {"directions": r["rays_d"].contiguous(), "i": r["i"], "j": r["j"], "inds": r["inds"]}        # all [1, H * W(, 3)]
        if len(_RAY_CACHE) >= _RAY_CACHE_MAX:
            _RAY_CACHE.pop(next(iter(_RAY_CACHE)))
        _RAY_CACHE[key] = hit
    B = poses.shape[0]
    d = hit["directions"] if B == 1 else hit["directions"].expand(B, -1, -1)
    rays_d = d @ poses[:, :3, :3].transpose(-1, -2)                                                            # utils.py:333
    rays_o = poses[..., :3, 3][..., None, :].expand_as(rays_d)                                                 # utils.py:335-336
    ex = (lambda t: t) if B == 1 else (lambda t: t.expand(B, -1))
    return {"i": ex(hit["i"]), "j": ex(hit["j"]), "inds": ex(hit["inds"]), "rays_o": rays_o, "rays_d": rays_d}


class TrainerMixin:
    """In front of the reference's `Trainer` (dropin/ernerf/nerf_triplane/utils.py): `test_gui_with_data` with the resize as one HIP launch and pinned copies."""
    _mf_pins = None
    _mf_linear_to_srgb = None          # set by the drop-in module to the reference's `linear_to_srgb`

    def _mf_pinned(self, H, W):
        ring = self.__dict__.get("_mf_pins")
        if ring is None or ring["shape"] != (H, W):
            ring = {"shape": (H, W), "k": 0,
                    "bufs": [(torch.empty(H, W, 3, dtype=torch.float32).pin_memory(), torch.empty(H, W, dtype=torch.float32).pin_memory()) for _ in range(4)]}
            self.__dict__["_mf_pins"] = ring
        ring["k"] = (ring["k"] + 1) % len(ring["bufs"])
        return ring["bufs"][ring["k"]]

    def test_gui_with_data(self, data, W, H):
        self.model.eval()
        if self.ema is not None:
            self.ema.store()
            self.ema.copy_to()
        with torch.no_grad():
            with torch.autocast("cuda", enabled=bool(self.fp16)):                                             # utils.py:1200 (torch.cuda.amp.autocast: the same context)
                preds, preds_depth = self.test_step(data, perturb=False)                                       # utils.py:1201-1204
        if self.ema is not None:
            self.ema.restore()
        if self.opt.color_space == 'linear':
            preds = type(self)._mf_linear_to_srgb(preds)                                                       # the reference's own function (utils.py:78-81)
        if not (torch.is_tensor(preds) and preds.is_cuda and preds.dtype == torch.float32 and preds_depth.dtype == torch.float32):
            return _reference_tail(preds, preds_depth, W, H)
        h, w = int(preds.shape[1]), int(preds.shape[2])
        img, dep = preds[0].contiguous(), preds_depth[0].contiguous()
        if (h, w) != (int(H), int(W)):                                                                         # utils.py:1208-1209 (F.interpolate at equal sizes is the identity)
            out_i = torch.empty(H, W, 3, device=img.device)
            out_d = torch.empty(H, W, device=img.device)
            _lib.check(_lib.lib().mf_nerf_resize_frame(C.c_void_p(img.data_ptr()), C.c_void_p(dep.data_ptr()), h, w, int(H), int(W), C.c_void_p(out_i.data_ptr()),
                                                       C.c_void_p(out_d.data_ptr()), None, C.c_void_p(torch.cuda.current_stream(img.device).cuda_stream)),
                       "mf_nerf_resize_frame")
            img, dep = out_i, out_d
        pin_i, pin_d = self._mf_pinned(int(H), int(W))
        pin_i.copy_(img, non_blocking=True)
        pin_d.copy_(dep, non_blocking=True)
        torch.cuda.current_stream(img.device).synchronize()
        return {'image': pin_i.numpy(), 'depth': pin_d.numpy()}


def _reference_tail(preds, preds_depth, W, H):
    """utils.py:1208-1221's operations, for results that are not fp32 device tensors (a CPU model, a caller-patched test_step)"""
    import torch.nn.functional as F
    preds = F.interpolate(preds.permute(0, 3, 1, 2), size=(H, W), mode='bilinear').permute(0, 2, 3, 1).contiguous()
    preds_depth = F.interpolate(preds_depth.unsqueeze(1), size=(H, W), mode='nearest').squeeze(1)
    return {'image': preds[0].detach().cpu().numpy(), 'depth': preds_depth[0].detach().cpu().numpy()}
