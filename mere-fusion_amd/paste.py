"""Paste-back of generated faces into the cached full frames on the GPU: thin host side of `mf_paste_frames`.

The reference does this per frame on the host after a D2H of every generated face (lipreal.py:207-214; musereal.py:238-247 +
musetalk/utils/blending.py:103-125).  Here the avatar's full frames, bboxes and masks are cached on the device once and a batch of
generated faces is composed into a batch of uint8 BGR frames in one launch; results are bit-exact with OpenCV's 8-bit arithmetic
(csrc/mf_blend.hip).  There is no CPU path."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _dev_u8(x, device):
    t = x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x))
    if t.dtype != torch.uint8:
        raise RuntimeError(f"expected a uint8 image, got {t.dtype}")
    return t.to(device).contiguous()


class AvatarFrames:
    """Device-resident cache of one avatar's full frames and paste geometry.

    frames      : [n, H, W, 3] uint8 BGR        (frame_list_cycle, lipreal.py:174-179 / musereal.py:169-179)
    bboxes      : n x (x1, y1, x2, y2)           MuseTalk order (coords.pkl, musereal.py:239); for Wav2Lip avatars pass `lip_order=True`
                                                 and the reference's (y1, y2, x1, x2) tuples (lipreal.py:208)
    masks       : n uint8 [h_i, w_i, 3] BGR images (mask_list_cycle) or None (Wav2Lip: rectangle copy)
    crop_boxes  : n x (x_s, y_s, x_e, y_e)       mask_coords_list_cycle
    """

    def __init__(self, frames, bboxes, masks=None, crop_boxes=None, lip_order=False, device="cuda"):
        if not torch.cuda.is_available():
            raise RuntimeError("AvatarFrames needs a HIP device; no CPU path exists here")
        self.device = torch.device(device)
        fr = frames if torch.is_tensor(frames) else torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(f) for f in frames])))
        self.frames = _dev_u8(fr, self.device)
        if self.frames.dim() != 4 or self.frames.shape[3] != 3:
            raise RuntimeError(f"frames must be [n, H, W, 3], got {tuple(self.frames.shape)}")
        self.n, self.H, self.W = self.frames.shape[0], self.frames.shape[1], self.frames.shape[2]
        self.bboxes = [(int(b[2]), int(b[0]), int(b[3]), int(b[1])) if lip_order else tuple(int(v) for v in b) for b in bboxes]
        if len(self.bboxes) != self.n:
            raise RuntimeError("one bbox per cached frame is required")
        self.masks = None if masks is None else [_dev_u8(m, self.device) for m in masks]
        self.crop_boxes = None if crop_boxes is None else [tuple(int(v) for v in c) for c in crop_boxes]
        if self.masks is not None:
            if self.crop_boxes is None or len(self.masks) != self.n or len(self.crop_boxes) != self.n:
                raise RuntimeError("masks need one crop box per cached frame")
            for m, (xs, ys, xe, ye) in zip(self.masks, self.crop_boxes):
                if tuple(m.shape) != (ye - ys, xe - xs, 3):
                    raise RuntimeError(f"mask shape {tuple(m.shape)} does not match its crop box {(xs, ys, xe, ye)}")

    def jobs(self, indices):
        arr = (_lib.MfPasteJob * len(indices))()
        for k, i in enumerate(indices):
            j = arr[k]
            j.frame_index = int(i)
            j.x1, j.y1, j.x2, j.y2 = self.bboxes[i]
            if self.masks is not None:
                j.cx1, j.cy1, j.cx2, j.cy2 = self.crop_boxes[i]
                j.mask = self.masks[i].data_ptr()
            else:
                j.mask = None
        return arr

    def paste(self, res, indices, out=None):
        """res: device [B, S, S', 3] uint8 (MuseTalk frames) or fp32 (Wav2Lip `pred * 255`); indices: the B cached-frame indices
        (mirror index of each result).  Returns uint8 [B, H, W, 3] on the device."""
        if not res.is_cuda:
            raise RuntimeError("paste needs the generated frames on the HIP device; no CPU path exists here")
        if res.dtype not in (torch.uint8, torch.float32) or res.dim() != 4 or res.shape[3] != 3 or res.shape[0] != len(indices):
            raise RuntimeError(f"res must be uint8 / float32 [B, h, w, 3] with B = {len(indices)}, got {res.dtype} {tuple(res.shape)}")
        res = res.contiguous()
        B = res.shape[0]
        if out is None:
            out = torch.empty((B, self.H, self.W, 3), dtype=torch.uint8, device=res.device)
        jobs = self.jobs(list(indices))
        with torch.cuda.device(res.device):
            _lib.check(_lib.lib().mf_paste_frames(res.data_ptr(), int(res.dtype == torch.float32), res.shape[1], res.shape[2], self.frames.data_ptr(),
                                                  self.n, self.H, self.W, jobs, B, out.data_ptr(),
                                                  C.c_void_p(torch.cuda.current_stream(res.device).cuda_stream)), "paste_frames")
        return out


def resize_linear_u8(src, dw, dh):
    """cv2.resize(src, (dw, dh)) (INTER_LINEAR) for a uint8 [h, w, 3] device tensor."""
    if not src.is_cuda or src.dtype != torch.uint8 or src.dim() != 3 or src.shape[2] != 3:
        raise RuntimeError("resize_linear_u8 needs a uint8 [h, w, 3] HIP device tensor; no CPU path exists here")
    src = src.contiguous()
    dst = torch.empty((dh, dw, 3), dtype=torch.uint8, device=src.device)
    with torch.cuda.device(src.device):
        _lib.check(_lib.lib().mf_resize_linear_u8(src.data_ptr(), src.shape[0], src.shape[1], dst.data_ptr(), int(dh), int(dw),
                                                  C.c_void_p(torch.cuda.current_stream(src.device).cuda_stream)), "resize_linear_u8")
    return dst
