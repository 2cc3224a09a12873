#!/usr/bin/env python3
"""Regenerates the tuning table shipped beside the library (mere-fusion_amd/tune/gfx950.txt) on an MI355X: every implicit-GEMM layer of the BASELINE.json
shapes measured once through the explicit warm-up API (mf_*_tune), appended to MF_TUNE_CACHE.

    MF_TUNE_CACHE=gpurun_out/gfx950_tune.txt python tools/make_tune_cache.py      # then copy the file to mere-fusion_amd/tune/gfx950.txt

Shapes: MuseTalk UNet + VAE at batch 8, 16, ... 64 (what MuseBatcher issues for 1 ... 8 sessions) and 1, 2 (tests / parity legs); Wav2Lip at batch 1, 2, 5, 16,
128; the same in the single-pass bf16 mode for the `alt` legs of bench.py."""
import os
import sys

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import torch

from mere_fusion_amd import weights as W
from mere_fusion_amd.musetalk.config import MUSETALK_V1, unet_config_json, vae_config_json
from mere_fusion_amd.musetalk.models.unet import UNet
from mere_fusion_amd.musetalk.models.vae import VAE
from mere_fusion_amd.wav2lip.models import Wav2Lip

assert os.environ.get("MF_TUNE_CACHE"), "set MF_TUNE_CACHE to the file the measurements are appended to"
open(os.environ["MF_TUNE_CACHE"], "a").close()
usd, vsd = W.make_musetalk_unet_state_dict(MUSETALK_V1, 0), W.make_musetalk_vae_state_dict(MUSETALK_V1, 0)
for prec, batches in (("bf16x3", (1, 2, 3, 5, 8, 16, 24, 32, 40, 48, 56, 64)), ("bf16", (1, 8))):
    unet = UNet(unet_config_json(MUSETALK_V1["unet"]), usd, precision=prec, max_batch=max(batches))
    vae = VAE(config=vae_config_json(MUSETALK_V1["vae"]), state_dict=vsd, precision=prec, max_batch=max(batches))
    for b in batches:
        lat, aud = W.make_musetalk_inputs(b, b)
        pred = unet.model(lat.cuda(), torch.tensor([0]).cuda(), encoder_hidden_states=unet.pe(aud.cuda())).sample
        vae.decode_latents_device(pred)
        unet.model.tune(b)
        vae.tune(b)
        print(f"musetalk {prec} batch {b}: tuned", flush=True)
    del unet, vae
    torch.cuda.empty_cache()
sd = W.make_wav2lip_state_dict(0)
for prec in ("bf16x3", "bf16"):
    m = Wav2Lip(precision=prec)
    m.load_state_dict(sd)
    m = m.to("cuda").eval()
    for b in (1, 2, 5, 16, 128):
        mel, face, _ = W.make_lip_inputs(b, b)
        with torch.no_grad():
            m(mel.cuda(), face.cuda())
        m.tune(b)
        print(f"wav2lip {prec} batch {b}: tuned", flush=True)
print(sum(1 for _ in open(os.environ["MF_TUNE_CACHE"])), "rows in", os.environ["MF_TUNE_CACHE"])
