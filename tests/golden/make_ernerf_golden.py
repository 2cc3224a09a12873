#!/usr/bin/env python3
"""Generates tests/golden/ernerf_golden.npz by running the REFERENCE's own Python (ernerf/nerf_triplane/network.py,
renderer.py, raymarching.py, grid.py, sphere_harmonics.py) on the CPU of the build container.

What is real and what is substituted:
  * real: `NeRFNetwork` / `NeRFRenderer` / `MLP` / `AudioNet` / `AudioAttNet` and the autograd wrappers -- imported from
    /root/reference, unmodified; the module tree, `forward`, `density`, `encode_x`, `encode_audio` and the inference branch of
    `run_cuda` (renderer.py:231-291) execute as written.
  * substituted: the four CUDA extensions (`_raymarching_face`, `_gridencoder`, `_shencoder`, `_freqencoder`), which need nvcc + an
    NVIDIA GPU.  Their entry points are backed here by the plain-C restatement oracle/ernerf_ref.c (so the goldens pin everything
    ABOVE the extension boundary to the reference; the kernels themselves stay "parity unpinned").
  * stubbed third-party imports the inference path never touches: trimesh, tensorboardX, cv2, mcubes, torch_ema, imageio, lpips.
  * `Tensor.cuda()` is the identity for this run (the wrappers call it unconditionally, raymarching.py:33-34).

    python tests/golden/make_ernerf_golden.py        # needs /root/reference; writes tests/golden/ernerf_golden.npz
"""
import argparse
import ctypes as C
import importlib
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
# The drop-in directory is on the path below only for the four extension-module NAMES the reference's wrappers import (their entry points are replaced by the
# C oracle right after).  Since round 6 it also shadows `ernerf.nerf_triplane.network` with the reference's class + the MI355X render path mixed in:
# MF_NERF_DROPIN=0 keeps that path out, so that what runs here -- and what the fixture records -- is the REFERENCE's own run_cuda.
os.environ["MF_NERF_DROPIN"] = "0"
os.environ["MF_PLACEMENT"] = "0"
sys.path[:0] = [ROOT, os.path.join(ROOT, "mere-fusion_amd", "dropin"), "/root/reference"]

from mere_fusion_amd import weights as W  # noqa: E402
from mere_fusion_amd.ernerf.field import grid_geometry  # noqa: E402

_p = lambda t: C.c_void_p(t.data_ptr())


def import_reference():
    for _ in range(60):
        try:
            return importlib.import_module("ernerf.nerf_triplane.network")
        except ModuleNotFoundError as e:
            sys.modules[e.name] = mock.MagicMock(name=e.name)
    raise RuntimeError("could not import the reference network")


def oracle_backends():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "libernerfref.so"))
    rm, ge, sh = types.SimpleNamespace(), types.SimpleNamespace(), types.SimpleNamespace()

    def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
        lib.ref_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), C.c_uint32(N), C.c_float(min_near), _p(nears), _p(fars))

    def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, dt_gamma, max_steps, Cc, H, grid, near, far, xyzs, dirs, deltas, noises):
        lib.ref_march_rays(C.c_uint32(n_alive), C.c_uint32(n_step), _p(rays_alive), _p(rays_t), _p(rays_o), _p(rays_d), C.c_float(bound),
                           C.c_float(dt_gamma), C.c_uint32(max_steps), C.c_uint32(Cc), C.c_uint32(H), _p(grid), _p(near), _p(far), _p(xyzs), _p(dirs),
                           _p(deltas), _p(noises))

    def composite_rays_triplane(n_alive, n_step, T_thresh, rays_alive, rays_t, sigmas, rgbs, deltas, aa, ae, unc, ws, depth, image, aas, aes, us):
        sigmas, rgbs, deltas, aa, ae, unc = (t.contiguous() for t in (sigmas, rgbs, deltas, aa, ae, unc))
        lib.ref_composite_rays_triplane(C.c_uint32(n_alive), C.c_uint32(n_step), C.c_float(T_thresh), _p(rays_alive), _p(rays_t), _p(sigmas), _p(rgbs),
                                        _p(deltas), _p(aa), _p(ae), _p(unc), _p(ws), _p(depth), _p(image), _p(aas), _p(aes), _p(us))

    def grid_encode_forward(inputs, embeddings, offsets, outputs, B, D, Cc, L, S, H, dy_dx, gridtype, align_corners):
        assert dy_dx is None
        lib.ref_grid_encode_forward(_p(inputs), _p(embeddings), _p(offsets), _p(outputs), C.c_uint32(B), C.c_uint32(D), C.c_uint32(Cc), C.c_uint32(L),
                                    C.c_float(S), C.c_uint32(H), C.c_uint32(gridtype), C.c_int(int(align_corners)))

    def sh_encode_forward(inputs, outputs, B, input_dim, degree, dy_dx):
        assert dy_dx is None and input_dim == 3
        lib.ref_sh_encode_forward(_p(inputs), _p(outputs), C.c_uint32(B), C.c_uint32(degree))

    rm.near_far_from_aabb, rm.march_rays, rm.composite_rays_triplane = near_far_from_aabb, march_rays, composite_rays_triplane
    ge.grid_encode_forward, sh.sh_encode_forward = grid_encode_forward, sh_encode_forward
    return rm, ge, sh


def build_reference_model(sd_field, seed, torso=False):
    net = import_reference()
    rm, ge, sh = oracle_backends()
    importlib.import_module("ernerf.raymarching.raymarching")._backend = rm
    importlib.import_module("ernerf.gridencoder.grid")._backend = ge
    importlib.import_module("ernerf.shencoder.sphere_harmonics")._backend = sh
    lib = C.CDLL(os.path.join(ROOT, "oracle", "libernerfref.so"))

    def freq_encode_forward(inputs, B, D, degree, Cc, outputs):
        lib.ref_freq_encode_forward(_p(inputs), C.c_uint32(B), C.c_uint32(D), C.c_uint32(degree), C.c_uint32(Cc), _p(outputs))
    importlib.import_module("ernerf.freqencoder.freq")._backend = types.SimpleNamespace(freq_encode_forward=freq_encode_forward)
    torch.Tensor.cuda = lambda self, *a, **k: self
    if torso:
        opt = argparse.Namespace(asr_model="esperanto", emb=False, att=2, bound=1, min_near=0.05, density_thresh=10, density_thresh_torso=0.01,
                                 exp_eye=True, test_train=False, smooth_lips=False, torso=True, cuda_ray=True, ind_num=4, ind_dim=4, ind_dim_torso=8,
                                 train_camera=False, unc_loss=1, torso_shrink=0.8)
        torch.manual_seed(seed)
        model = net.NeRFNetwork(opt).eval()
        own = model.state_dict()
        for k, v in sd_field.items():
            assert own[k].shape == v.shape, (k, own[k].shape, v.shape)
        model.load_state_dict(sd_field, strict=False)
        return model, None
    opt = argparse.Namespace(asr_model="esperanto", emb=False, att=2, bound=1, min_near=0.05, density_thresh=10, density_thresh_torso=0.01,
                             exp_eye=True, test_train=False, smooth_lips=False, torso=False, cuda_ray=True, ind_num=16, ind_dim=4,
                             train_camera=False, unc_loss=1)
    torch.manual_seed(seed)
    model = net.NeRFNetwork(opt).eval()
    own = model.state_dict()
    for k, v in sd_field.items():
        assert own[k].shape == v.shape, (k, own[k].shape, v.shape)
    # the audio nets keep torch's seeded init (their tensors go into the fixture); the field takes the repository's seeded weights
    audio = W.make_ernerf_audio_state_dict(own, seed)
    model.load_state_dict({**sd_field, **audio}, strict=False)
    model.testing = True
    return model, audio


def main():
    seed = 0
    offsets, pls = grid_geometry()
    sd = W.make_ernerf_field_state_dict(int(offsets[-1]), seed)
    sd = {k: (v * 0.35 if k.startswith("sigma_net.net.2") else v) for k, v in sd.items()}
    model, audio_sd = build_reference_model(sd, seed)
    assert np.array_equal(model.encoder_xy.offsets.numpy(), offsets)
    out = {"offsets": offsets, "log2_per_level_scale": np.float32(np.log2(model.encoder_xy.per_level_scale))}
    g = torch.Generator().manual_seed(seed)
    # ---- a20: NeRFNetwork.forward on 400 samples ----
    M = 400
    x = (torch.rand(M, 3, generator=g) * 2 - 1) * torch.tensor([1.0, 0.5, 1.0])
    d = torch.randn(M, 3, generator=g); d = d / d.norm(dim=1, keepdim=True)
    enc_a = torch.randn(1, 32, generator=g)
    c = model.individual_codes[0:1].detach().clone()
    e = torch.tensor([[0.4]])
    with torch.no_grad():
        sigma, color, aa, ae, unc = model(x, d, enc_a, c[0], e)
    out.update(field_x=x.numpy(), field_d=d.numpy(), field_enc_a=enc_a.numpy(), field_c=c.numpy(), field_e=e.numpy(), field_sigma=sigma.numpy(),
               field_color=color.numpy(), field_amb_aud=aa.numpy(), field_amb_eye=ae.numpy(), field_unc_first=unc.reshape(-1)[:M].numpy(),
               field_unc_shape=np.array(unc.shape))
    # ---- a23: encode_audio on an [8, 44, 16] window ----
    auds = torch.randn(8, 44, 16, generator=g)
    with torch.no_grad():
        enc_audio = model.encode_audio(auds)
    out.update(auds=auds.numpy(), enc_audio=enc_audio.numpy())
    for k, v in audio_sd.items():
        out["audio_sd/" + k] = v.numpy()
    # ---- a15: run_cuda (inference branch) on 24 x 24 rays through the ball ----
    Wd = 24
    bitfield = W.make_ernerf_sphere_bitfield()
    model.density_bitfield.copy_(torch.from_numpy(bitfield))
    model.density_scale = 40.0
    ro, rd = W.make_ernerf_camera_rays(Wd)
    bg = torch.tensor([0.1, 0.2, 0.3]).expand(Wd * Wd, 3).contiguous()
    with torch.no_grad():
        res = model.render(torch.from_numpy(ro)[None], torch.from_numpy(rd)[None], auds, torch.zeros(1, Wd * Wd, 2), torch.eye(4)[None], eye=e,
                           index=[0], staged=True, bg_color=bg, perturb=False, dt_gamma=1 / 256, max_steps=16, T_thresh=1e-4)
    out.update(render_W=np.int32(Wd), render_image=res["image"].reshape(-1, 3).numpy(), render_depth=res["depth"].reshape(-1).numpy(),
               render_amb_aud=res["ambient_aud"].reshape(-1).numpy(), render_amb_eye=res["ambient_eye"].reshape(-1).numpy(),
               render_ind_code=model.individual_codes[0].detach().numpy())
    # ---- a22: run_torso / forward_torso on 40 x 40 background pixels, reference model built with opt.torso ----
    from mere_fusion_amd.ernerf.field import grid_geometry as gg
    t_offsets, t_pls = gg(num_levels=16, base_resolution=16, log2_hashmap_size=16, desired_resolution=2048)
    tsd = W.make_ernerf_torso_state_dict(int(t_offsets[-1]), seed)
    tmodel, _ = build_reference_model(tsd, seed, torso=True)
    assert np.array_equal(tmodel.torso_encoder.offsets.numpy(), t_offsets)
    Wt = 40
    u = (torch.arange(Wt, dtype=torch.float32) + 0.5) / Wt * 2 - 1
    yy, xx = torch.meshgrid(u, u, indexing="ij")
    bg_coords = torch.stack([xx, yy], -1).reshape(1, -1, 2).contiguous()
    pose = torch.eye(4); pose[:3, :3] = torch.tensor([[0.98, 0.05, -0.19], [-0.03, 0.995, 0.09], [0.195, -0.083, 0.977]]); pose[:3, 3] = torch.tensor([0.05, -0.02, 0.9])
    tbg = torch.tensor([0.3, 0.5, 0.7]).expand(Wt * Wt, 3).contiguous()
    with torch.no_grad():
        tr = tmodel.run_torso(torch.zeros(1, Wt * Wt, 3), bg_coords, pose[None], 0, tbg)
    out.update(torso_W=np.int32(Wt), torso_bg_coords=bg_coords.reshape(-1, 2).numpy(), torso_pose=pose.numpy(), torso_bg_in=tbg.numpy(),
               torso_bg_color=tr["bg_color"].numpy(), torso_alpha=tr["torso_alpha"].numpy(), torso_offsets=t_offsets,
               torso_log2_per_level_scale=np.float32(np.log2(tmodel.torso_encoder.per_level_scale)))
    path = os.path.join(ROOT, "tests", "golden", "ernerf_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: getattr(v, "shape", None) for k, v in out.items() if not k.startswith("audio_sd/")})


if __name__ == "__main__":
    main()
