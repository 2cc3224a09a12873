#!/usr/bin/env python3
"""Measures the launch configurations of the MuseTalk UNet + VAE (and optionally Wav2Lip) at the given batch sizes into MF_TUNE_CACHE (MF_DEBUG=tune prints
every layer's model pick against the measured winner): the quick form of tools/make_tune_cache.py for kernel work on one batch size.

    MF_TUNE_CACHE=gpurun_out/tune_b8.txt MF_DEBUG=tune python tools/tune_one_batch.py 8 [16 ...] [--wav2lip 16]
"""
import os
import sys

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import torch

from mere_fusion_amd import weights as W
from mere_fusion_amd.musetalk.config import MUSETALK_V1, unet_config_json, vae_config_json
from mere_fusion_amd.musetalk.models.unet import UNet
from mere_fusion_amd.musetalk.models.vae import VAE

assert os.environ.get("MF_TUNE_CACHE"), "set MF_TUNE_CACHE to the file the measurements are appended to"
open(os.environ["MF_TUNE_CACHE"], "a").close()
args = sys.argv[1:]
w2l = []
if "--wav2lip" in args:
    i = args.index("--wav2lip")
    w2l = [int(x) for x in args[i + 1:]]
    args = args[:i]
batches = [int(x) for x in args] or [8]
if batches != [0]:
    usd, vsd = W.make_musetalk_unet_state_dict(MUSETALK_V1, 0), W.make_musetalk_vae_state_dict(MUSETALK_V1, 0)
    unet = UNet(unet_config_json(MUSETALK_V1["unet"]), usd, precision="bf16x3", max_batch=max(batches))
    vae = VAE(config=vae_config_json(MUSETALK_V1["vae"]), state_dict=vsd, precision="bf16x3", max_batch=max(batches))
    for b in batches:
        lat, aud = W.make_musetalk_inputs(b, b)
        pred = unet.model(lat.cuda(), torch.tensor([0]).cuda(), encoder_hidden_states=unet.pe(aud.cuda())).sample
        vae.decode_latents_device(pred)
        unet.model.tune(b)
        vae.tune(b)
        print(f"musetalk bf16x3 batch {b}: tuned", flush=True)
if w2l:
    from mere_fusion_amd.wav2lip.models import Wav2Lip
    m = Wav2Lip(precision="bf16x3")
    m.load_state_dict(W.make_wav2lip_state_dict(0))
    m = m.to("cuda").eval()
    for b in w2l:
        mel, face, _ = W.make_lip_inputs(b, b)
        with torch.no_grad():
            m(mel.cuda(), face.cuda())
        m.tune(b)
        print(f"wav2lip bf16x3 batch {b}: tuned", flush=True)
print(sum(1 for _ in open(os.environ["MF_TUNE_CACHE"])), "rows in", os.environ["MF_TUNE_CACHE"])
