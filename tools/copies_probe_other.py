#!/usr/bin/env python3
"""Batch-position independence of the other networks: Wav2Lip (16 identical inputs), the VAE decoder at 64 frames and the UNet at 16 / 40 / 64 identical frames."""
import os, sys
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import torch
from mere_fusion_amd import weights as W
from mere_fusion_amd.musetalk.config import MUSETALK_V1, unet_config_json, vae_config_json
from mere_fusion_amd.musetalk.models.unet import UNet
from mere_fusion_amd.musetalk.models.vae import VAE
from mere_fusion_amd.wav2lip.models import Wav2Lip
def report(name, out):
    d = [(out[k].float() - out[0].float()).abs().max().item() for k in range(out.shape[0])]
    print(f"{name}: max over copies {max(d):.3e}; copies that differ from copy 0: {[k for k, v in enumerate(d) if v > 0][:20]}", flush=True)
m = Wav2Lip(); m.load_state_dict(W.make_wav2lip_state_dict(0)); m = m.cuda().eval()
g = torch.Generator().manual_seed(0)
mel, face = torch.randn(1, 1, 80, 16, generator=g), torch.rand(1, 6, 96, 96, generator=g)
for B in (16, 5, 128):
    with torch.no_grad():
        for call in range(2):
            report(f"wav2lip B={B} call {call}", m(mel.repeat(B, 1, 1, 1).cuda(), face.repeat(B, 1, 1, 1).cuda()))
usd, vsd = W.make_musetalk_unet_state_dict(MUSETALK_V1, 0), W.make_musetalk_vae_state_dict(MUSETALK_V1, 0)
unet = UNet(unet_config_json(MUSETALK_V1["unet"]), usd, max_batch=64)
vae = VAE(config=vae_config_json(MUSETALK_V1["vae"]), state_dict=vsd, max_batch=64)
lat, aud = W.make_musetalk_inputs(1, 3)
for B in (16, 40, 64, 8, 3):
    pred = unet.model(lat.repeat(B, 1, 1, 1).cuda(), torch.tensor([0]).cuda(), encoder_hidden_states=unet.pe(aud.repeat(B, 1, 1).cuda())).sample
    report(f"unet B={B}", pred)
    report(f"vae  B={B}", vae.decode_latents_device(pred))
