"""The ER-NeRF drop-in seam (VERDICT r05 missing #1): `from ernerf.nerf_triplane.network import NeRFNetwork` (app.py:17) with only PYTHONPATH changed must give
the reference's own class with the MI355X render path in front, and a frame rendered through `model.render(...)` must come from the device loop
(mf_nerf_head_render) and match the reference's golden frame.

CPU part (needs the reference checkout, which the GPU box does not have): the import chain, the class wiring, invalidation on load_state_dict, the
no-CPU-path error.  GPU part (no reference there): the same mixin in front of a stand-in base that carries the reference's attributes and state-dict names --
the golden frame of tests/golden/ernerf_golden.npz was rendered by the reference's NeRFNetwork itself (tests/golden/make_ernerf_golden.py)."""
import argparse
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
REF = "/root/reference"

_PROBE = r'''
import sys, importlib, argparse
from unittest import mock
for _ in range(60):
    try:
        net = importlib.import_module("ernerf.nerf_triplane.network")      # app.py:17
        break
    except ModuleNotFoundError as e:                                          # third-party modules the inference path never touches (cv2, trimesh, ...)
        sys.modules[e.name] = mock.MagicMock(name=e.name)
import torch
from mere_fusion_amd.ernerf.network import HipRenderMixin
mro = net.NeRFNetwork.__mro__
assert mro[1] is HipRenderMixin, mro
assert mro[2].__module__ == "ernerf.nerf_triplane._reference_network" and mro[2].__name__ == "NeRFNetwork", mro
assert net._ref.__file__.startswith("REFROOT"), net._ref.__file__
assert net.AudioNet is net._ref.AudioNet and net.MLP is net._ref.MLP
import ernerf.nerf_triplane.renderer as R, ernerf.nerf_triplane.provider as P    # fall through to the reference's own files
assert R.__file__.startswith("REFROOT") and P.__file__.startswith("REFROOT")
assert issubclass(net.NeRFNetwork, R.NeRFRenderer)
opt = argparse.Namespace(asr_model="esperanto", emb=False, att=2, bound=1, min_near=0.05, density_thresh=10, density_thresh_torso=0.01, exp_eye=True,
                         test_train=False, smooth_lips=False, torso=False, cuda_ray=True, ind_num=16, ind_dim=4, train_camera=False, unc_loss=1)
m = net.NeRFNetwork(opt).eval()                                                  # app.py:379
keys = set(m.state_dict())
assert {"encoder_xy.embeddings", "sigma_net.net.0.weight", "audio_net.encoder_conv.0.weight", "density_bitfield", "individual_codes"} <= keys
m.__dict__["_mf"] = {"stale": True}
m.load_state_dict(m.state_dict())                                                # Trainer.load_checkpoint
assert m._mf is None, "load_state_dict must drop the device objects"
m.__dict__["_mf"] = {"stale": True}
m.float()
assert m._mf is None, "_apply must drop the device objects"
N = 16
try:
    m.render(torch.zeros(1, N, 3), torch.zeros(1, N, 3), torch.zeros(8, 44, 16), torch.zeros(1, N, 2), torch.eye(4)[None], eye=torch.zeros(1, 1), index=[0],
             staged=True, bg_color=None, perturb=False, dt_gamma=1 / 256, max_steps=16)
    raise SystemExit("rendering CPU tensors must raise")
except RuntimeError as e:
    assert "no CPU path" in str(e), e
# ---- utils: app.py:16 `from ernerf.nerf_triplane.utils import *`, provider.py `from .utils import get_rays` ------------------------------------------------
import numpy as np
from mere_fusion_amd.ernerf.frontend import TrainerMixin
U = importlib.import_module("ernerf.nerf_triplane.utils")
assert not U.__file__.startswith("REFROOT") and U._ref.__file__.startswith("REFROOT"), (U.__file__, U._ref.__file__)
assert U.Trainer.__mro__[1] is TrainerMixin and U.Trainer.__mro__[2] is U._ref.Trainer, U.Trainer.__mro__
assert P.get_rays is U.get_rays and R.custom_meshgrid is U._ref.custom_meshgrid                       # the reference's own modules bind the drop-in's names
missing = [n for n in vars(U._ref) if not n.startswith("_") and n not in vars(U)]
assert not missing, missing                                                                              # `import *` exports what the reference's module exports
g = torch.Generator().manual_seed(5)
def pose(n):
    q, _ = torch.linalg.qr(torch.randn(n, 3, 3, generator=g))
    m = torch.eye(4).repeat(n, 1, 1); m[:, :3, :3] = q; m[:, :3, 3] = torch.randn(n, 3, generator=g)
    return m
intr = np.array([1200.0, 1190.0, 8.3, 5.9])
for n in (1, 1, 2):                                                                                      # first call fills the cache, the second one hits it
    ps = pose(n)
    want = U._ref.get_rays(ps, intr, 12, 17, -1)
    got = U.get_rays(ps, intr, 12, 17, -1)
    for k in ("rays_o", "rays_d", "i", "j", "inds"):
        assert got[k].shape == want[k].shape and torch.equal(got[k], want[k]), (n, k)
assert U.get_rays(pose(1), intr, 12, 17, 7)["rays_d"].shape == (1, 7, 3)                                 # sampled rays: the reference's function
# Trainer.test_gui_with_data on CPU tensors falls back to the reference's operations: same arrays
class _Model:
    def eval(self): pass
tr = U.Trainer.__new__(U.Trainer)
tr.model, tr.ema, tr.fp16, tr.opt = _Model(), None, False, argparse.Namespace(color_space="linear")
img, dep = torch.rand(1, 9, 11, 3, generator=g), torch.rand(1, 9, 11, generator=g)
tr.test_step = lambda data, perturb=False: (img.clone(), dep.clone())
want = U._ref.Trainer.test_gui_with_data(tr, {}, 14, 10)
got = tr.test_gui_with_data({}, 14, 10)
assert got["image"].shape == (10, 14, 3) and np.array_equal(got["image"], want["image"]) and np.array_equal(got["depth"], want["depth"])
print("DROPIN-OK")
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "ernerf")), reason="no reference checkout here (the GPU box)")
def test_reference_import_resolves_to_the_mixed_in_class(lib_built):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "mere-fusion_amd", "dropin"), REF]), MF_PLACEMENT="0")
    out = subprocess.run([sys.executable, "-c", _PROBE.replace("REFROOT", REF)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "DROPIN-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


class _ReferenceShapedBase(torch.nn.Module):
    """What `HipRenderMixin` touches of the reference's NeRFNetwork / NeRFRenderer (renderer.py:62-133, network.py:93-165): attributes, buffers and parameters
    under the reference's names.  Its own `run_cuda` fails: if the mixin fell back to the base, the test would know."""

    def __init__(self, opt, sd):
        super().__init__()
        self.opt, self.bound, self.grid_size, self.density_scale, self.min_near = opt, opt.bound, 128, 1, opt.min_near
        self.exp_eye, self.test_train, self.smooth_lips, self.torso, self.train_camera = opt.exp_eye, False, opt.smooth_lips, False, False
        self.individual_dim, self.emb, self.att = opt.ind_dim, False, opt.att
        self.density_thresh_torso, self.mean_density_torso = 0.01, 0.0
        self.individual_codes = torch.nn.Parameter(torch.zeros(opt.ind_num, opt.ind_dim))
        self.register_buffer("density_bitfield", torch.zeros(128 ** 3 // 8, dtype=torch.uint8))
        self._names = {}
        for k, v in sd.items():
            name = "p_" + k.replace(".", "__")
            self.register_parameter(name, torch.nn.Parameter(v.clone(), requires_grad=False))
            self._names[name] = k
        if self.smooth_lips:
            self.enc_a = None

    def state_dict(self, *a, **k):
        sd = super().state_dict(*a, **k)
        return {self._names.get(key, key): v for key, v in sd.items()}

    def load_state_dict(self, sd, *a, **k):
        back = {v: n for n, v in self._names.items()}
        return super().load_state_dict({back.get(key, key): v for key, v in sd.items()}, *a, **k)

    def run_cuda(self, *a, **k):
        raise AssertionError("the reference's run_cuda was reached: the device loop did not run")

    def render(self, rays_o, rays_d, auds, bg_coords, poses, staged=False, max_ray_batch=4096, **kwargs):          # renderer.py:657-677
        return self.run_cuda(rays_o, rays_d, auds, bg_coords, poses, **kwargs)


@pytest.mark.gpu
@pytest.mark.parametrize("smooth", [False, True])
def test_hip_render_through_the_mixin_matches_the_reference_golden(lib_built, smooth):
    from mere_fusion_amd import weights as W
    from mere_fusion_amd.ernerf.network import HipRenderMixin
    g = np.load(os.path.join(ROOT, "tests", "golden", "ernerf_golden.npz"))
    sd = W.make_ernerf_field_state_dict(int(g["offsets"][-1]), 0)
    sd = {k: (v * 0.35 if k.startswith("sigma_net.net.2") else v) for k, v in sd.items()}
    sd.update({k[len("audio_sd/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("audio_sd/")})
    opt = argparse.Namespace(asr_model="esperanto", emb=False, att=2, bound=1, min_near=0.05, exp_eye=True, smooth_lips=smooth, ind_num=16, ind_dim=4)

    class Net(HipRenderMixin, _ReferenceShapedBase):
        pass

    m = Net(opt, sd)
    with torch.no_grad():
        m.individual_codes[0].copy_(torch.from_numpy(g["render_ind_code"]))
        m.density_bitfield.copy_(torch.from_numpy(W.make_ernerf_sphere_bitfield()))
    m = m.cuda().eval()
    m.density_scale = 40.0                                                      # set after construction, as the GUI does: read per frame
    Wd = int(g["render_W"])
    ro, rd = W.make_ernerf_camera_rays(Wd)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    bg = torch.tensor([0.1, 0.2, 0.3]).expand(Wd * Wd, 3).contiguous().cuda()
    kw = dict(eye=cu(g["field_e"]), index=[0], staged=True, bg_color=bg, perturb=False, dt_gamma=1 / 256, max_steps=16, T_thresh=1e-4)
    res = m.render(cu(ro)[None], cu(rd)[None], cu(g["auds"]), torch.zeros(1, Wd * Wd, 2, device="cuda"), torch.eye(4, device="cuda")[None], **kw)
    assert m.mf_frames == 1 and tuple(res["image"].shape) == (1, Wd * Wd, 3) and tuple(res["depth"].shape) == (1, Wd * Wd)
    err = np.abs(res["image"].reshape(-1, 3).cpu().numpy() - g["render_image"]).max()
    derr = np.abs(res["depth"].reshape(-1).cpu().numpy() - g["render_depth"]).max()
    aerr = np.abs(res["ambient_aud"].reshape(-1).cpu().numpy() - g["render_amb_aud"])
    eerr = np.abs(res["ambient_eye"].reshape(-1).cpu().numpy() - g["render_amb_eye"])
    print(f"model.render through the drop-in mixin (smooth_lips={smooth}) vs the reference's golden frame: image {err:.3e}, depth {derr:.3e}, "
          f"ambient sums {aerr.max():.3e} / {eerr.max():.3e}")
    assert err <= 1e-3 and derr <= 1e-3
    assert (aerr / (1 + np.abs(g["render_amb_aud"]))).max() <= 2e-3 and eerr.max() <= 2e-3
    # a second frame: the first frame's lazy sums are gone, the EMA state lives on the module as the reference keeps it (renderer.py:190-194)
    res2 = m.render(cu(ro)[None], cu(rd)[None], cu(g["auds"]) * 1.1, torch.zeros(1, Wd * Wd, 2, device="cuda"), torch.eye(4, device="cuda")[None], **kw)
    assert m.mf_frames == 2
    m.render(cu(ro)[None], cu(rd)[None], cu(g["auds"]), torch.zeros(1, Wd * Wd, 2, device="cuda"), torch.eye(4, device="cuda")[None], **kw)
    with pytest.raises(KeyError, match="gone"):
        res2["ambient_aud"]                                                    # res2's frame is no longer the head's last one
    if smooth:
        assert m.enc_a is not None and tuple(m.enc_a.shape) == (1, 32)
        assert float((res2["image"] - res["image"]).abs().max()) > 0            # another audio window, another frame
    # weights change -> the device objects are rebuilt from the new state dict
    first = m._mf["renderer"]
    m.load_state_dict(m.state_dict())
    assert m._mf is None
    m.render(cu(ro)[None], cu(rd)[None], cu(g["auds"]), torch.zeros(1, Wd * Wd, 2, device="cuda"), torch.eye(4, device="cuda")[None], **kw)
    assert m._mf["renderer"] is not first
    # training mode leaves the fast path (and here reaches the stand-in's failing run_cuda)
    m.train()
    with pytest.raises(AssertionError, match="reference's run_cuda"):
        m.render(cu(ro)[None], cu(rd)[None], cu(g["auds"]), torch.zeros(1, Wd * Wd, 2, device="cuda"), torch.eye(4, device="cuda")[None], **kw)


@pytest.mark.gpu
def test_hip_render_through_the_mixin_with_the_torso_branch(lib_built):
    """`opt.torso` models (the deployed ER-NeRF checkpoints): `run_cuda` mixes the head over `run_torso`'s background (renderer.py:272-277).  The mixin must build the
    torso from the module's own tensors (`torso_*`, `anchor_points`, `individual_codes_torso`, `density_grid_torso`, `mean_density_torso` as set AFTER the load) and give
    the same frame as the hand-assembled HipHeadRenderer + HipTorso + HipAudioEncoder that tests/test_ernerf.py holds to the reference's goldens piece by piece."""
    from mere_fusion_amd import weights as W
    from mere_fusion_amd.ernerf.audio import HipAudioEncoder
    from mere_fusion_amd.ernerf.field import HipNeRFField, grid_geometry
    from mere_fusion_amd.ernerf.network import HipRenderMixin
    from mere_fusion_amd.ernerf.renderer import HipHeadRenderer
    from mere_fusion_amd.ernerf.torso import HipTorso
    g = np.load(os.path.join(ROOT, "tests", "golden", "ernerf_golden.npz"))
    sd = W.make_ernerf_field_state_dict(int(g["offsets"][-1]), 0)
    sd = {k: (v * 0.35 if k.startswith("sigma_net.net.2") else v) for k, v in sd.items()}
    sd.update({k[len("audio_sd/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("audio_sd/")})
    t_offsets, _ = grid_geometry(num_levels=16, base_resolution=16, log2_hashmap_size=16, desired_resolution=2048)
    tsd = W.make_ernerf_torso_state_dict(int(t_offsets[-1]), 0)
    opt = argparse.Namespace(asr_model="esperanto", emb=False, att=2, bound=1, min_near=0.05, exp_eye=True, smooth_lips=True, ind_num=16, ind_dim=4, torso_shrink=0.8)

    class Base(_ReferenceShapedBase):
        def __init__(self, opt, sd, tsd):
            super().__init__(opt, {**sd, **{k: v for k, v in tsd.items() if k not in ("density_grid_torso",)}})
            self.torso, self.individual_dim_torso = True, 8
            self.register_buffer("density_grid_torso", tsd["density_grid_torso"].clone())

    class Net(HipRenderMixin, Base):
        pass

    m = Net(opt, sd, tsd)
    with torch.no_grad():
        m.individual_codes[0].copy_(torch.from_numpy(g["render_ind_code"]))
        m.density_bitfield.copy_(torch.from_numpy(W.make_ernerf_sphere_bitfield()))
    m = m.cuda().eval()
    m.density_scale, m.mean_density_torso = 40.0, 0.3                       # `load_checkpoint` assigns mean_density_torso after the state dict (utils.py)
    Wd = 48
    ro, rd = W.make_ernerf_camera_rays(Wd)
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    u = (torch.arange(Wd, dtype=torch.float32) + 0.5) / Wd * 2 - 1
    yy, xx = torch.meshgrid(u, u, indexing="ij")
    bg_coords = torch.stack([xx, yy], -1).reshape(1, -1, 2).contiguous().cuda()
    pose = torch.from_numpy(g["torso_pose"])[None].cuda()
    bg = torch.tensor([0.3, 0.5, 0.7]).expand(Wd * Wd, 3).contiguous().cuda()
    auds = [cu(g["auds"]), cu(g["auds"]) * 1.2]
    kw = dict(eye=cu(g["field_e"]), index=[0], staged=True, bg_color=bg, perturb=False, dt_gamma=1 / 256, max_steps=16, T_thresh=1e-4)
    got = [m.render(cu(ro)[None], cu(rd)[None], a, bg_coords, pose, **kw) for a in auds]
    assert m.mf_frames == 2
    # the same two frames from hand-assembled objects
    full = {k: v.detach() for k, v in m.state_dict().items()}
    r = HipHeadRenderer(HipNeRFField(full, max_samples=Wd * Wd), m.density_bitfield, density_scale=40.0,
                        torso=HipTorso(full, torso_shrink=0.8, individual_dim=8, density_thresh_torso=0.01, mean_density_torso=0.3, max_pixels=Wd * Wd),
                        audio=HipAudioEncoder(full, att=2), ind_code=m.individual_codes[0].detach(), smooth_lips=True)
    for i, a in enumerate(auds):
        want = r.render(cu(ro), cu(rd), a, bg_coords, pose, cu(g["field_e"]), bg_color=bg, loop="device", dt_gamma=1 / 256, max_steps=16, T_thresh=1e-4)
        assert torch.equal(got[i]["image"].reshape(-1, 3), want["image"]) and torch.equal(got[i]["depth"].reshape(-1), want["depth"]), i
    assert float((got[0]["image"] - got[1]["image"]).abs().max()) > 0          # another audio window (and the EMA carried on the module): another frame
    img = got[0]["image"].reshape(Wd, Wd, 3)
    assert float(img.std()) > 0.02 and float((img - bg.reshape(Wd, Wd, 3)).abs().amax(-1).gt(1e-3).float().mean()) > 0.05   # head and torso both drew something


@pytest.mark.gpu
def test_frontend_rays_and_gui_tail_on_the_gpu(lib_built):
    """The two per-frame pieces either side of `model.render` (mere_fusion_amd/ernerf/frontend.py) on the device: `get_rays` against the operations of
    utils.py:274-336 (whole-frame branch, restated here: the GPU box has no reference checkout) -- same bits, first call and cache hit, one and two poses; the
    `test_gui_with_data` mixin against F.interpolate + .cpu() (utils.py:1208-1212) at equal and at different sizes, and the pinned result ring (an array stays
    what it was for the next three frames)."""
    import torch.nn.functional as F
    from mere_fusion_amd.ernerf import frontend as fe

    def ref_get_rays(poses, intrinsics, H, W, N=-1, patch_size=1, rect=None):
        device = poses.device
        B = poses.shape[0]
        fx, fy, cx, cy = intrinsics
        i, j = torch.meshgrid(torch.linspace(0, W - 1, W, device=device), torch.linspace(0, H - 1, H, device=device), indexing="ij")
        i = i.t().reshape([1, H * W]).expand([B, H * W]) + 0.5
        j = j.t().reshape([1, H * W]).expand([B, H * W]) + 0.5
        inds = torch.arange(H * W, device=device).expand([B, H * W])
        zs = torch.ones_like(i)
        xs = (i - cx) / fx * zs
        ys = (j - cy) / fy * zs
        directions = torch.stack((xs, ys, zs), dim=-1)
        directions = directions / torch.norm(directions, dim=-1, keepdim=True)
        rays_d = directions @ poses[:, :3, :3].transpose(-1, -2)
        rays_o = poses[..., :3, 3][..., None, :].expand_as(rays_d)
        return {"i": i, "j": j, "inds": inds, "rays_o": rays_o, "rays_d": rays_d}
    g = torch.Generator().manual_seed(9)

    def pose(n):
        q, _ = torch.linalg.qr(torch.randn(n, 3, 3, generator=g))
        m = torch.eye(4).repeat(n, 1, 1)
        m[:, :3, :3] = q
        m[:, :3, 3] = torch.randn(n, 3, generator=g)
        return m.cuda()
    intr = np.array([1200.0, 1190.0, 225.3, 221.9])
    for n in (1, 1, 2):
        ps = pose(n)
        want, got = ref_get_rays(ps, intr, 450, 450), fe.get_rays(ref_get_rays, ps, intr, 450, 450)
        for k in ("rays_o", "rays_d", "i", "j", "inds"):
            assert got[k].shape == want[k].shape and torch.equal(got[k], want[k]), (n, k)

    class _Model:
        def eval(self):
            pass

    class _T(fe.TrainerMixin):
        _mf_linear_to_srgb = staticmethod(lambda x: torch.where(x < 0.0031308, 12.92 * x, 1.055 * x ** 0.41666 - 0.055))

        def __init__(self, color_space):
            self.model, self.ema, self.fp16, self.opt = _Model(), None, True, argparse.Namespace(color_space=color_space)
            self.frames = []

        def test_step(self, data, perturb=False):
            return self.frames[-1]
    for (h, w, H, W), cs in (((24, 24, 24, 24), "srgb"), ((24, 32, 45, 70), "linear"), ((50, 50, 450, 450), "srgb")):
        t = _T(cs)
        outs = []
        for k in range(5):
            img, dep = torch.rand(1, h, w, 3, generator=g).cuda(), torch.rand(1, h, w, generator=g).cuda()
            t.frames.append((img, dep))
            out = t.test_gui_with_data({}, W, H)
            pi = t._mf_linear_to_srgb(img) if cs == "linear" else img
            want = F.interpolate(pi.permute(0, 3, 1, 2), size=(H, W), mode="bilinear").permute(0, 2, 3, 1).contiguous()[0].cpu().numpy()
            want_d = F.interpolate(dep.unsqueeze(1), size=(H, W), mode="nearest").squeeze(1)[0].cpu().numpy()
            assert out["image"].dtype == np.float32 and out["image"].shape == (H, W, 3) and out["depth"].shape == (H, W)
            assert np.abs(out["image"] - want).max() <= 2.4e-7 and np.array_equal(out["depth"], want_d), (h, w, H, W, k)
            outs.append((out["image"], want))
            for a, b in outs[-4:]:                                   # the ring: the last four results are still what they were
                assert np.abs(a - b).max() <= 2.4e-7
