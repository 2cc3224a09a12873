"""`ernerf.nerf_triplane.network.NeRFNetwork` as app.py:17 imports it, with the inference branch of its render path on MI355X (VERDICT r05 missing #1).

The reference builds the model (`NeRFNetwork(opt)`, app.py:379), lets its `Trainer` load the checkpoint into it (utils.py `load_checkpoint` ->
`model.load_state_dict`), and then calls `model.render(...)` once per frame from `Trainer.test_gui_with_data` (utils.py:1190-1223 -> `test_step` :926-956).
None of those files change.  `HipRenderMixin` goes IN FRONT of the reference's class (`dropin/ernerf/nerf_triplane/network.py`):

    class NeRFNetwork(HipRenderMixin, <the reference's NeRFNetwork>): pass

so the object is the reference's module in every respect -- same parameters, state-dict keys, `opt`, training methods -- except that `run_cuda`
(renderer.py:158-291), when called the way inference calls it, runs as one enqueue of HIP kernels:

    encode_audio + the lip-smoothing EMA (network.py:222-237, renderer.py:187-194)   mf_audio_encoder_forward_smooth
    run_torso (renderer.py:294-352, network.py:166-201)                              mf_nerf_torso_forward
    near/far, <= max_steps rounds of march -> field -> composite, background mix     mf_nerf_head_render (round control on the device, no host sync)

The device objects are built from `self.state_dict()` at the first frame AFTER the checkpoint is in (and again after any later `load_state_dict` / `.to()` /
`.half()`), i.e. with the weights the reference would have used for that frame (an EMA shadow copied in by `test_gui_with_data` included).  Training mode,
camera optimisation at test time, `perturb`, a missing audio window or eye feature, CPU tensors: the reference's own `run_cuda` runs, over the extension
shims (`dropin/_raymarching_face.py` ...), exactly as before."""
import os

import torch

from .. import _lib, placement


class _Results(dict):
    """`run_cuda`'s result dict.  `image` and `depth` are what `test_step` reads (utils.py:953-954); the three per-ray sums the reference also returns
    (renderer.py:286-288) are copied out of the head's buffers only if somebody asks for them -- before the next frame is rendered."""

    def __init__(self, owner, frame_id, prefix, n, *a, **k):
        super().__init__(*a, **k)
        self._owner, self._frame_id, self._prefix, self._n = owner, frame_id, prefix, n

    def __missing__(self, key):
        if key not in ("ambient_aud", "ambient_eye", "uncertainty"):
            raise KeyError(key)
        o = self._owner
        if o._mf_frame_id != self._frame_id:
            raise KeyError(f"{key}: the sums of this frame are gone (a later frame has been rendered); read them before the next render()")
        dev = self["image"].device
        aa, ae, un = (torch.empty(self._n, device=dev) for _ in range(3))
        import ctypes as C
        _lib.check(_lib.lib().mf_nerf_head_sums(o._mf["renderer"]._head, self._n, C.c_void_p(aa.data_ptr()), C.c_void_p(ae.data_ptr()),
                                                C.c_void_p(un.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "mf_nerf_head_sums")
        self["ambient_aud"], self["ambient_eye"], self["uncertainty"] = aa.view(*self._prefix), ae.view(*self._prefix), un
        return self[key]


class HipRenderMixin:
    _mf = None
    _mf_frame_id = 0
    mf_frames = 0                      # frames rendered by the device loop (tests assert the fast route ran)

    def __init__(self, *args, **kwargs):
        # on a multi-GPU node the process takes its GPU here, before the reference's Trainer says `.to('cuda')` (placement.py)
        placement.charge_session(self)
        super().__init__(*args, **kwargs)

    # ---- anything that changes the weights (or where they live) drops the device objects ------------------------------------------
    def _mf_drop(self):
        self.__dict__["_mf"] = None

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._mf_drop()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._mf_drop()
        return out

    def _mf_build(self, n_rays, device):
        from .audio import HipAudioEncoder
        from .field import HipNeRFField
        from .renderer import HipHeadRenderer
        from .torso import HipTorso
        opt = self.opt
        with torch.no_grad():
            sd = {k: v.detach() for k, v in self.state_dict().items()}
        cap = max(int(n_rays), 1)
        field = HipNeRFField(sd, bound=self.bound, individual_dim=self.individual_dim, exp_eye=self.exp_eye, max_samples=cap, device=device)
        torso = None
        if self.torso:
            torso = HipTorso(sd, torso_shrink=getattr(opt, "torso_shrink", 0.8), individual_dim=self.individual_dim_torso,
                             density_thresh_torso=self.density_thresh_torso, mean_density_torso=self.mean_density_torso, grid_size=self.grid_size,
                             max_pixels=cap, device=device)
        audio = None if self.emb else HipAudioEncoder(sd, att=self.att, device=device)
        rend = HipHeadRenderer(field, self.density_bitfield, bound=self.bound, min_near=self.min_near, density_scale=self.density_scale,
                               grid_size=self.grid_size, torso=torso, audio=audio, ind_code=None, smooth_lips=self.smooth_lips)
        self.__dict__["_mf"] = {"renderer": rend, "cap": cap, "device": device, "bitfield_ptr": self.density_bitfield.data_ptr()}
        return self._mf

    def _mf_fast_path(self, rays_o, auds, eye, perturb):
        if os.environ.get("MF_NERF_DROPIN", "1") == "0" or self.training or perturb:
            return False
        if self.train_camera and self.test_train:                     # renderer.py:170-175: a per-index camera offset, training-time machinery
            return False
        if auds is None or (self.exp_eye and eye is None):             # the reference's own error / None handling applies
            return False
        return True

    def run_cuda(self, rays_o, rays_d, auds, bg_coords, poses, eye=None, index=0, dt_gamma=0, bg_color=None, perturb=False, force_all_rays=False,
                 max_steps=1024, T_thresh=1e-4, **kwargs):
        if not self._mf_fast_path(rays_o, auds, eye, perturb):
            return super().run_cuda(rays_o, rays_d, auds, bg_coords, poses, eye=eye, index=index, dt_gamma=dt_gamma, bg_color=bg_color, perturb=perturb,
                                    force_all_rays=force_all_rays, max_steps=max_steps, T_thresh=T_thresh, **kwargs)
        if not (torch.is_tensor(rays_o) and rays_o.is_cuda):
            raise RuntimeError("NeRFNetwork.run_cuda (MI355X drop-in): rays must be HIP device tensors; there is no CPU path (MF_NERF_DROPIN=0 runs the "
                               "reference's loop over the extension shims, which need a device too)")
        prefix = rays_o.shape[:-1]
        n = int(prefix.numel())
        dev = rays_o.device
        st = self._mf
        if st is None or st["cap"] < n or st["device"] != dev or st["bitfield_ptr"] != self.density_bitfield.data_ptr():
            st = self._mf_build(n, dev)
        r = st["renderer"]
        # per-frame attributes the reference reads at call time (renderer.py:197-202, 259, 325): they may have been set after construction (`load_checkpoint`
        # assigns mean_density_torso, the GUI density_scale)
        r.bitfield, r.density_scale = self.density_bitfield, float(self.density_scale)
        r.ind_code = self.individual_codes[0].detach() if self.individual_dim > 0 else None
        if r.torso is not None:
            r.torso.thresh = float(min(self.density_thresh_torso, self.mean_density_torso))
        with torch.no_grad():
            if self.emb:                                                # network.py:229-230: an embedding lookup in front of the audio net -- the reference's modules run it
                enc_a = super().encode_audio(auds)
                if enc_a is not None and self.smooth_lips:
                    if self.enc_a is not None:
                        enc_a = 0.35 * self.enc_a + (1 - 0.35) * enc_a
                    self.enc_a = enc_a
                auds_in, r.smooth_lips = enc_a, False
            else:
                auds_in, r.smooth_lips = auds, bool(self.smooth_lips)
                r.enc_a = self.enc_a if self.smooth_lips else None
            out = r.render(rays_o, rays_d, auds_in, bg_coords, poses, eye, bg_color=bg_color, loop="device", dt_gamma=float(dt_gamma),
                           max_steps=int(max_steps), T_thresh=float(T_thresh))
            if self.smooth_lips and not self.emb:
                self.enc_a = r.enc_a                                    # the EMA state lives where the reference keeps it (renderer.py:128-129, 190-194)
        self.__dict__["_mf_frame_id"] = self._mf_frame_id + 1
        self.__dict__["mf_frames"] = self.mf_frames + 1
        res = _Results(self, self._mf_frame_id, prefix, n)
        res["depth"] = out["depth"].view(*prefix)
        res["image"] = out["image"].view(*prefix, 3)
        res["weights_sum"] = out["weights_sum"]
        return res


def load_reference_module(shadow_name, shadow_file, package):
    """The module the drop-in shadows, loaded from the NEXT directory of its package path under a private name, with the package set so that its relative
    imports (`from .renderer import NeRFRenderer`, `from ..encoding import get_encoder`) resolve through the same (extended) package."""
    import importlib
    import importlib.util
    import sys
    pkg = importlib.import_module(package)
    base = os.path.basename(shadow_file)
    for d in pkg.__path__:
        f = os.path.join(d, base)
        if os.path.exists(f) and os.path.abspath(f) != os.path.abspath(shadow_file):
            name = package + "._reference_" + os.path.splitext(base)[0]
            if name in sys.modules:
                return sys.modules[name]
            spec = importlib.util.spec_from_file_location(name, f)
            mod = importlib.util.module_from_spec(spec)
            mod.__package__ = package
            sys.modules[name] = mod
            try:
                spec.loader.exec_module(mod)
            except BaseException:
                sys.modules.pop(name, None)
                raise
            return mod
    raise ImportError(f"{shadow_name}: the reference's own {package.replace('.', '/')}/{base} was not found behind the drop-in -- put the mere-fusion checkout on "
                      "PYTHONPATH after mere-fusion_amd/dropin (INTEGRATION.md section 5)")
