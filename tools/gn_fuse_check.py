#!/usr/bin/env python3
"""Frames of the full VAE decoder at batch 8 (+ pre-clamp image) to a .npz: run once per build / environment and compare (the fused GroupNorm path must reproduce
the k_affine_silu_to_q path bit for bit: same operations on the same values).   python tools/gn_fuse_check.py out.npz [other.npz to compare with]"""
import os, sys
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np
import torch
from mere_fusion_amd import weights as W
from mere_fusion_amd.musetalk.config import MUSETALK_V1, vae_config_json
from mere_fusion_amd.musetalk.models.vae import VAE
vae = VAE(config=vae_config_json(MUSETALK_V1["vae"]), state_dict=W.make_musetalk_vae_state_dict(MUSETALK_V1, 0), max_batch=8)
lat = (torch.randn(8, 4, 32, 32, generator=torch.Generator().manual_seed(5)) * 0.9).cuda()
frames, image = vae.decode_latents_device(lat, want_image=True)
f2, _ = vae.decode_latents_device(lat, want_image=True)             # (second call: graph capture; third: replay)
f3, _ = vae.decode_latents_device(lat, want_image=True)
assert torch.equal(frames, f2) and torch.equal(frames, f3)
np.savez(sys.argv[1], frames=frames.cpu().numpy(), image=image.cpu().numpy())
print("frames std", float(frames.float().std()))
if len(sys.argv) > 2:
    o = np.load(sys.argv[2])
    d = np.abs(frames.cpu().numpy().astype(int) - o["frames"].astype(int))
    print("vs", sys.argv[2], ": uint8 max diff", d.max(), "differing", float((d > 0).mean()), "image max abs diff", float(np.abs(image.cpu().numpy() - o["image"]).max()))
