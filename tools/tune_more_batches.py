#!/usr/bin/env python3
"""Adds MuseTalk step sizes to a tuning table (GPU box): MF_TUNE_CACHE=gpurun_out/tune_more.txt python tools/tune_more_batches.py 72 80
(rows are appended to MF_TUNE_CACHE; append them to mere-fusion_amd/tune/gfx950.txt afterwards)."""
import os
import sys

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import torch

from mere_fusion_amd import weights as W
from mere_fusion_amd.musetalk.config import MUSETALK_V1, unet_config_json, vae_config_json
from mere_fusion_amd.musetalk.models.unet import UNet
from mere_fusion_amd.musetalk.models.vae import VAE

assert os.environ.get("MF_TUNE_CACHE"), "set MF_TUNE_CACHE to the file the measurements are appended to"
batches = [int(a) for a in sys.argv[1:]] or [72, 80]
open(os.environ["MF_TUNE_CACHE"], "a").close()
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
print(sum(1 for _ in open(os.environ["MF_TUNE_CACHE"])), "rows in", os.environ["MF_TUNE_CACHE"])
