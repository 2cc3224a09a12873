"""ORACLE (test infrastructure, not product): CPU restatement of the inference branch of `NeRFRenderer.run_cuda`
(ernerf/nerf_triplane/renderer.py:231-291) over the plain-C kernel restatements (oracle/ernerf_ref.c) and the torch field
restatement (oracle/ernerf_net_ref.py).  PINNED above the extension boundary: the fixture of tests/golden/make_ernerf_golden.py holds a
24 x 24 frame rendered by the reference's own `NeRFRenderer.run_cuda`; this loop reproduces it to 2e-6.  Only tests/, smoke() and bench.py's cpu_baseline
leg may import this module."""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import ernerf_net_ref as NR

HERE = os.path.dirname(os.path.abspath(__file__))
_p = lambda a: a.ctypes.data_as(C.c_void_p)
F = lambda *s: np.zeros(s, np.float32)


def run_cuda(sd, offsets, S, rays_o, rays_d, enc_a, ind_code, eye, bitfield, bound=1.0, min_near=0.05, bg_color=1.0, dt_gamma=1 / 256,
             max_steps=16, T_thresh=1e-4, H=128, field=None, density_scale=1.0):
    lib = C.CDLL(os.path.join(HERE, "libernerfref.so"))
    ro = np.ascontiguousarray(rays_o, np.float32).reshape(-1, 3); rd = np.ascontiguousarray(rays_d, np.float32).reshape(-1, 3)
    N = ro.shape[0]
    cascade = 1 + math.ceil(math.log2(bound))
    aabb = np.array([-bound, -bound / 2, -bound, bound, bound / 2, bound], np.float32)
    nears, fars = F(N), F(N)
    lib.ref_near_far_from_aabb(_p(ro), _p(rd), _p(aabb), C.c_uint32(N), C.c_float(min_near), _p(nears), _p(fars))
    weights_sum, depth, image = F(N), F(N), F(N, 3)
    aa_sum, ae_sum, un_sum = F(N), F(N), F(N)
    alive = np.arange(N, dtype=np.int32)
    rays_t = nears.copy()
    step, trace = 0, []
    field = field or (lambda x, d: NR.field_forward(sd, x, d, enc_a, ind_code, eye, offsets, S, bound=bound))
    while step < max_steps:
        n_alive = alive.shape[0]
        if n_alive <= 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        M = n_alive * n_step
        xyzs, dirs, deltas = F(M, 3), F(M, 3), F(M, 2)
        noises = F(n_alive)
        lib.ref_march_rays(C.c_uint32(n_alive), C.c_uint32(n_step), _p(alive), _p(rays_t), _p(ro), _p(rd), C.c_float(bound), C.c_float(dt_gamma),
                           C.c_uint32(max_steps), C.c_uint32(cascade), C.c_uint32(H), _p(bitfield), _p(nears), _p(fars), _p(xyzs), _p(dirs),
                           _p(deltas), _p(noises))
        sig, rgb, aa, ae, un = field(torch.from_numpy(xyzs), torch.from_numpy(dirs))
        sig, rgb = np.ascontiguousarray((density_scale * sig).numpy(), np.float32), np.ascontiguousarray(rgb.numpy(), np.float32)   # renderer.py:261
        aa, ae, un = (np.ascontiguousarray(t.numpy().reshape(-1), np.float32) for t in (aa, ae, un))
        alive = alive.copy()
        lib.ref_composite_rays_triplane(C.c_uint32(n_alive), C.c_uint32(n_step), C.c_float(T_thresh), _p(alive), _p(rays_t), _p(sig), _p(rgb),
                                        _p(deltas), _p(aa), _p(ae), _p(un), _p(weights_sum), _p(depth), _p(image), _p(aa_sum), _p(ae_sum), _p(un_sum))
        alive = alive[alive >= 0]
        trace.append((n_alive, n_step))
        step += n_step
    bg = np.broadcast_to(np.asarray(bg_color, np.float32), (N, 3)) if np.ndim(bg_color) else np.float32(bg_color)
    img = np.clip(image + (1 - weights_sum)[:, None] * bg, 0, 1).astype(np.float32)            # renderer.py:275-277
    dep = np.maximum(depth - nears, 0) / (fars - nears)                                       # renderer.py:279
    return {"image": img, "depth": dep.astype(np.float32), "ambient_aud": aa_sum, "ambient_eye": ae_sum, "weights_sum": weights_sum, "trace": trace,
            "frame_u8": (img * 255).astype(np.uint8)}


def resize_frame(image, depth, H, W):
    """`Trainer.test_gui_with_data` tail (ernerf/nerf_triplane/utils.py:1208-1216) + nerfreal.py:111 in numpy fp32: image [h, w, 3] bilinear
    with half-pixel centres (aten upsample_bilinear2d, align_corners=False), depth [h, w] nearest (legacy floor rule), frame = uint8(image * 255).
    Pinned: tests/test_ernerf.py checks it against torch.nn.functional.interpolate -- the very call the reference makes -- on CPU."""
    image = np.asarray(image, np.float32)
    depth = np.asarray(depth, np.float32)
    h, w = depth.shape
    f = np.float32

    def src(n_out, n_in):
        s = f(n_in) / f(n_out)
        # aten's `scale * (dst + 0.5) - 0.5` is compiled to one fused multiply-add (CPU and CUDA builds alike): a float product is exact in
        # double, so rounding once from double reproduces it
        x = (np.float64(s) * (np.arange(n_out, dtype=np.float32) + f(0.5)).astype(np.float64) - 0.5).astype(np.float32)
        x = np.maximum(x, f(0))
        i0 = x.astype(np.int64)
        i1 = i0 + (i0 < n_in - 1)
        l1 = (x - i0.astype(np.float32)).astype(np.float32)
        return i0, i1, (f(1) - l1).astype(np.float32), l1
    y0, y1, ly0, ly1 = src(H, h)
    x0, x1, lx0, lx1 = src(W, w)
    lx0, lx1 = lx0[None, :, None], lx1[None, :, None]
    top = lx0 * image[y0][:, x0] + lx1 * image[y0][:, x1]
    bot = lx0 * image[y1][:, x0] + lx1 * image[y1][:, x1]
    out = (ly0[:, None, None] * top + ly1[:, None, None] * bot).astype(np.float32)
    ny = np.minimum(np.floor(np.arange(H, dtype=np.float32) * (f(h) / f(H))).astype(np.int64), h - 1)
    nx = np.minimum(np.floor(np.arange(W, dtype=np.float32) * (f(w) / f(W))).astype(np.int64), w - 1)
    return out, depth[ny][:, nx], (out * f(255)).astype(np.uint8)
