"""The head render loop of ER-NeRF on MI355X: the inference branch of `NeRFRenderer.run_cuda`
(ernerf/nerf_triplane/renderer.py:158-291) over the HIP kernels, one frame per call.

The reference's own renderer keeps working unchanged on top of the extension shims (dropin/_raymarching_face.py ...) and a
`model.forward = HipNeRFField(...).forward` hook; this module is the same loop without the torch.autograd wrappers, used by
bench.py / smoke() / the tests, and it is what `nerfreal.py:render` amounts to per frame once torso and audio nets are fixed.
The compaction `rays_alive[rays_alive >= 0]` (renderer.py:266) and its host sync are kept as in the reference."""
import ctypes as C
import os

import torch

from .. import _lib
from . import _raymarching_face as rm


class HipHeadRenderer:
    def __init__(self, field, density_bitfield, bound=1.0, min_near=0.05, density_scale=1.0, grid_size=128, torso=None, audio=None,
                 ind_code=None, smooth_lips=False):
        import math
        self.field = field
        self.torso, self.audio, self.ind_code, self.smooth_lips, self.enc_a = torso, audio, ind_code, bool(smooth_lips), None
        self.bound, self.min_near, self.density_scale, self.grid_size = float(bound), float(min_near), float(density_scale), int(grid_size)
        self.cascade = 1 + math.ceil(math.log2(bound))                                   # renderer.py:69
        self.bitfield = density_bitfield.contiguous()
        b = self.bound
        self.aabb_infer = torch.tensor([-b, -b / 2, -b, b, b / 2, b], dtype=torch.float32, device=density_bitfield.device)   # renderer.py:86-89
        self._lib = _lib.lib()
        self._head, self._side = None, None

    @torch.no_grad()
    def render(self, rays_o, rays_d, auds, bg_coords, poses, eye, bg_color=None, **kw):
        """One frame as `NeRFRenderer.render` -> `run_cuda` does it (renderer.py:657-677, 158-291): audio window -> enc_a (+ the lip
        smoothing EMA of :190-194), fixed individual code 0 (:197-202), head loop, torso / background mix (:272-277).
        loop="device" runs the head with device-side round control."""
        def audio_part():
            if self.audio is not None and self.smooth_lips and hasattr(self.audio, "encode_audio_smooth"):
                # the EMA of renderer.py:190-194 inside the encoder's launch (three torch elementwise launches per frame otherwise; same bits)
                enc_a = self.audio.encode_audio_smooth(auds, self.enc_a)
                if enc_a is not None:
                    self.enc_a = enc_a
                return enc_a
            enc_a = self.audio.encode_audio(auds) if self.audio is not None else auds
            if enc_a is not None and self.smooth_lips:
                if self.enc_a is not None:
                    enc_a = 0.35 * self.enc_a + (1 - 0.35) * enc_a
                self.enc_a = enc_a
            return enc_a
        device_loop = kw.pop("loop", "host") == "device"
        # (the torso on a second stream beside the head paid while it was a ~0.3 ms launch chain; as one 71 us kernel the cross-stream dependency cost more
        # than the overlap returned, 0.737 vs 0.711 ms per frame: the path left in round 5)
        enc_a = audio_part()
        if device_loop and self.torso is not None and not kw.get("graph"):
            # the head loop is enqueued FIRST and left unfinished; the torso's per-frame host work (the wrapped anchors: a 4 x 4 inverse and 42 sines, ~90 us, and the
            # device -> host copy of the pose when it lives on the device as the reference's does) then runs while the GPU marches, and the background mix closes the
            # frame.  Same kernels, same bits; a caller that syncs every frame (the reference's loop, through the drop-in) no longer pays that host time on top
            out = self.run_cuda_device(rays_o, rays_d, enc_a, self.ind_code, eye, bg_color=None, finish=False, **kw)
            return self.finish_device(out, self.torso.run_torso(bg_coords, poses, bg_color)["bg_color"])
        if self.torso is not None:
            bg_color = self.torso.run_torso(bg_coords, poses, bg_color)["bg_color"]
        if device_loop:
            return self.run_cuda_device(rays_o, rays_d, enc_a, self.ind_code, eye, bg_color=bg_color, **kw)
        return self.run_cuda(rays_o, rays_d, enc_a, self.ind_code, eye, bg_color=bg_color, **kw)

    @torch.no_grad()
    def resize(self, out, h, w, H, W, want_u8=True):
        """`test_gui_with_data`'s resize to the GUI size (utils.py:1208-1216): image bilinear, depth nearest, uint8 frame of nerfreal.py:111."""
        dev = out["image"].device
        img, dep = out["image"].contiguous(), out["depth"].contiguous()
        assert img.numel() == 3 * h * w and dep.numel() == h * w
        res = {"image": torch.empty(H, W, 3, device=dev), "depth": torch.empty(H, W, device=dev),
               "frame_u8": torch.empty(H, W, 3, dtype=torch.uint8, device=dev) if want_u8 else None}
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(self._lib.mf_nerf_resize_frame(p(img), p(dep), h, w, H, W, p(res["image"]), p(res["depth"]), p(res["frame_u8"]),
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)), "mf_nerf_resize_frame")
        return res

    def finish_device(self, out, bg_color):
        """Background mix / depth normalisation / uint8 frame for a run_cuda_device(finish=False) result."""
        N = out["image"].shape[0]
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        bg = bg_color.float().contiguous() if torch.is_tensor(bg_color) else None
        per_ray = bg is not None and bg.numel() == 3 * N
        bgc = float(1.0 if bg_color is None else (0.0 if bg is not None else bg_color))
        _lib.check(self._lib.mf_nerf_head_finish(self._head, N, p(bg), int(per_ray), bgc, p(out["image"]), p(out["depth"]), p(out["weights_sum"]),
                                                 p(out["frame_u8"]), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "mf_nerf_head_finish")
        return out

    def __del__(self):
        h = getattr(self, "_head", None)
        if h:
            self._lib.mf_nerf_head_destroy(h)
            self._head = None

    @torch.no_grad()
    def run_cuda_device(self, rays_o, rays_d, enc_a, ind_code, eye, bg_color=None, dt_gamma=1 / 256, max_steps=16, T_thresh=1e-4, want_u8=False,
                        graph=False, finish=True):
        """The same frame as run_cuda with the round control on the device (mf_nerf_head_render): one enqueue, no host sync between
        rounds.  graph=True captures the enqueue once per (ray count, tensors, launched rounds) into a CUDA graph and replays it."""
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N, dev = rays_o.shape[0], rays_o.device
        if getattr(self, "_head", None) is None or self._head_cap < N:
            if getattr(self, "_head", None):
                self._lib.mf_nerf_head_destroy(self._head)
            self._head = C.c_void_p()
            _lib.check(self._lib.mf_nerf_head_create(self.field._h, N, C.byref(self._head)), "mf_nerf_head_create")
            self._head_cap, self._graphs = N, {}
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        bg = bg_color.float().contiguous() if torch.is_tensor(bg_color) else None
        per_ray = bg is not None and bg.numel() == 3 * N
        bgc = float(1.0 if bg_color is None else (0.0 if bg is not None else bg_color))
        ea = enc_a.float().reshape(-1).contiguous()
        ic = ind_code.float().reshape(-1).contiguous() if ind_code is not None else None
        # the eye feature: a device tensor (the reference's loader hands one over, provider.py) is read by the kernels where it lives -- no device -> host copy, no
        # sync in front of the frame; a host tensor / None goes by value as before
        eye_dev = eye.float().reshape(-1)[:1].contiguous() if (torch.is_tensor(eye) and eye.is_cuda) else None
        ev = 0.0 if (eye is None or eye_dev is not None) else float(eye.reshape(-1)[0])
        self._eye_keep = eye_dev
        _lib.check(self._lib.mf_nerf_head_set_eye(self._head, p(eye_dev)), "mf_nerf_head_set_eye")

        def enqueue(out):
            _lib.check(self._lib.mf_nerf_head_render(self._head, p(rays_o), p(rays_d), N, p(self.bitfield), self.cascade, self.grid_size, self.min_near,
                                                     float(dt_gamma), int(max_steps), float(T_thresh), self.density_scale, p(ea), p(ic), ev, p(bg), int(per_ray) if finish else -1,
                                                     bgc, p(out["image"]), p(out["depth"]), p(out["weights_sum"]), p(out["frame_u8"]),
                                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)), "mf_nerf_head_render")

        def fresh():
            return {"image": torch.empty(N, 3, device=dev), "depth": torch.empty(N, device=dev), "weights_sum": torch.empty(N, device=dev),
                    "frame_u8": torch.empty(N, 3, dtype=torch.uint8, device=dev) if want_u8 else None}
        if not graph:
            out = fresh()
            enqueue(out)
            return out
        # graph mode: static input / output buffers per (ray count, scalar arguments); inputs are copied in unless they already live there.  The number of rounds
        # enqueued as launches (the tail launch stands for the rest) follows the frames before: a graph is captured per count -- two or three in a session
        planned = C.c_int(0)
        _lib.check(self._lib.mf_nerf_head_plan_rounds(self._head, int(max_steps), C.byref(planned)), "mf_nerf_head_plan_rounds")
        key = (N, ev, eye_dev is not None, bool(want_u8), bgc, bool(finish), None if bg is None else bg.numel(), float(dt_gamma), int(max_steps), float(T_thresh),
               planned.value)
        hit = self._graphs.get(key)
        live = (rays_o, rays_d, ea, ic, bg, eye_dev)
        if hit is None:
            static = tuple(None if t is None else t.clone() for t in live)
            rays_o, rays_d, ea, ic, bg, eye_dev = static
            _lib.check(self._lib.mf_nerf_head_set_eye(self._head, p(eye_dev)), "mf_nerf_head_set_eye")
            out = fresh()
            _lib.check(self._lib.mf_nerf_head_set_rounds(self._head, planned.value), "mf_nerf_head_set_rounds")
            try:
                enqueue(out)                               # warm-up outside the capture
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    enqueue(out)
            finally:
                _lib.check(self._lib.mf_nerf_head_set_rounds(self._head, -1), "mf_nerf_head_set_rounds")
            hit = (g, out, static)
            self._graphs[key] = hit
        else:
            for dst, src in zip(hit[2], live):
                if dst is not None and dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
        hit[0].replay()
        return hit[1]

    @torch.no_grad()
    def run_cuda(self, rays_o, rays_d, enc_a, ind_code, eye, bg_color=None, dt_gamma=1 / 256, max_steps=16, T_thresh=1e-4, perturb=False,
                 want_u8=False):
        """rays_o / rays_d: [N, 3] (or [1, N, 3]) CUDA fp32.  Returns image [N, 3] in [0, 1], depth [N], the ambient / weight sums and,
        with want_u8, the uint8 frame of nerfreal.py:111."""
        rays_o = rays_o.contiguous().view(-1, 3).float()
        rays_d = rays_d.contiguous().view(-1, 3).float()
        N, dev = rays_o.shape[0], rays_o.device
        nears, fars = torch.empty(N, device=dev), torch.empty(N, device=dev)
        rm.near_far_from_aabb(rays_o, rays_d, self.aabb_infer, N, self.min_near, nears, fars)
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        weights_sum, depth, image = z(N), z(N), z(N, 3)
        amb_aud_sum, amb_eye_sum, unc_sum = z(N), z(N), z(N)
        rays_alive = torch.arange(N, dtype=torch.int32, device=dev)
        rays_t = nears.clone()
        step = 0
        trace = []
        while step < max_steps:
            n_alive = rays_alive.shape[0]
            if n_alive <= 0:
                break
            n_step = max(min(N // n_alive, 8), 1)                                        # renderer.py:256
            M = n_alive * n_step
            xyzs, dirs, deltas = z(M, 3), z(M, 3), z(M, 2)                                # raymarching.py:383-385
            noises = torch.rand(n_alive, device=dev) if (perturb and step == 0) else z(n_alive)
            rm.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, self.bound, dt_gamma, max_steps, self.cascade, self.grid_size,
                          self.bitfield, nears, fars, xyzs, dirs, deltas, noises)
            sigmas, rgbs, amb_aud, amb_eye, unc = self.field.forward(xyzs, dirs, enc_a, ind_code, eye)
            if self.density_scale != 1.0:
                sigmas = self.density_scale * sigmas
            rm.composite_rays_triplane(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, amb_aud.view(-1), amb_eye.view(-1),
                                       unc.view(-1), weights_sum, depth, image, amb_aud_sum, amb_eye_sum, unc_sum)
            rays_alive = rays_alive[rays_alive >= 0]                                     # renderer.py:266
            trace.append((n_alive, n_step))
            step += n_step
        frame = torch.empty(N, 3, dtype=torch.uint8, device=dev) if want_u8 else None
        per_ray = torch.is_tensor(bg_color) and bg_color.numel() == 3 * N
        bg = bg_color.float().contiguous() if torch.is_tensor(bg_color) else None
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(self._lib.mf_nerf_finish(p(image), p(depth), p(weights_sum), p(nears), p(fars), p(bg), int(per_ray),
                                            float(1.0 if bg_color is None else (0.0 if bg is not None else bg_color)), N, p(frame),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "mf_nerf_finish")
        return {"image": image, "depth": depth, "ambient_aud": amb_aud_sum, "ambient_eye": amb_eye_sum, "uncertainty": unc_sum,
                "weights_sum": weights_sum, "frame_u8": frame, "trace": trace}
