#!/usr/bin/env python3
"""Image / uint8 error of the VAE decoder against the fp32 oracle under channel-scale stress (tests/test_musetalk_stress.py), for a list of scale ranges.
    python tools/vae_stress_probe.py            (f16 + FP6 resnet convs)      MF_CONV_Q=0 python tools/vae_stress_probe.py   (bf16x3 everywhere)"""
import os, sys
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np
import torch
from mere_fusion_amd import weights as W
from mere_fusion_amd.musetalk.config import MUSETALK_V1, vae_config_json
from mere_fusion_amd.musetalk.models.vae import VAE
from oracle import musetalk_ref as R
from tests.test_musetalk_stress import stressed_vae_state_dict

torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
vsd0 = W.make_musetalk_vae_state_dict(MUSETALK_V1, 0)
lat = (torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(41)) * 0.18215).repeat(4, 1, 1, 1)
print("MF_CONV_Q =", os.environ.get("MF_CONV_Q", "1"))
for lo, hi, frac, gain, one in ((1.0, 1.0, 0.0, 1.0, False), (0.1, 10.0, 0.01, 30.0, False), (1e-2, 1e2, 0.01, 30.0, False), (0.1, 10.0, 0.01, 30.0, True), (1e-2, 1e2, 0.01, 30.0, True)):
    vsd, _ = stressed_vae_state_dict(vsd0, 1, lo, hi, frac, gain, one)
    vae = VAE(config=vae_config_json(MUSETALK_V1["vae"]), state_dict=vsd, max_batch=8)
    want_img = R.vae_decode(vsd, MUSETALK_V1["vae"], lat[:2] / MUSETALK_V1["vae"]["scaling_factor"])
    want_u8 = R.decode_latents(vsd, MUSETALK_V1["vae"], lat[:2])
    frames, image = vae.decode_latents_device(lat.cuda(), want_image=True)
    ierr = (image.cpu()[:2] - want_img).abs().max().item()
    d = np.abs(frames.cpu().numpy()[:2].astype(int) - want_u8.astype(int))
    print(f"{'one-sided (weights only)' if one else 're-parametrised'} scales [{lo:g}, {hi:g}], outliers {frac:g} x {gain:g}: image L-inf {ierr:.3e} (|image| <= {float(want_img.abs().max()):.2f}), uint8 max {d.max()}, differing {100 * (d > 0).mean():.3f} %", flush=True)
    del vae
    torch.cuda.empty_cache()
