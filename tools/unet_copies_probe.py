#!/usr/bin/env python3
"""Runs the full-size MuseTalk UNet on a batch of IDENTICAL frames and reports how the outputs of the copies differ (they must not): per copy the largest
difference to copy 0 and where it sits in the 32 x 32 latent -- the first thing to look at when a kernel change makes results depend on the batch position.
    python tools/unet_copies_probe.py [batch] [calls]        (MF_LIB_PATH=build_ab/lib<variant>.so: another build of the library, tools/pkfma_variants.sh)"""
import os, sys
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import torch
from mere_fusion_amd import weights as W
from mere_fusion_amd.musetalk.config import MUSETALK_V1, unet_config_json
from mere_fusion_amd.musetalk.models.unet import UNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
usd = W.make_musetalk_unet_state_dict(MUSETALK_V1, 0)
unet = UNet(unet_config_json(MUSETALK_V1["unet"]), usd, max_batch=B)
lat, aud = W.make_musetalk_inputs(1, 3)
lat, aud = lat.repeat(B, 1, 1, 1).cuda(), aud.repeat(B, 1, 1).cuda()
bad_calls, worst, first = 0, 0.0, None
for rep in range(calls):
    out = unet.model(lat, torch.tensor([0]).cuda(), encoder_hidden_states=unet.pe(aud)).sample.float().cpu()
    if first is None:
        first = out
    d = (out - out[:1]).abs()
    per = d.flatten(1).max(1).values
    across = (out - first).abs().max().item()          # call-to-call: the same inputs must give the same bits every time
    if per.max() > 0 or across > 0:
        bad_calls += 1
        worst = max(worst, per.max().item(), across)
    if rep < 3 or (per.max() > 0 and bad_calls <= 5):
        print(f"call {rep}: per-copy max diff vs copy 0:", [f"{v:.1e}" for v in per.tolist()], f"vs call 0: {across:.1e}")
        if per.max() > 0:
            b = int(per.argmax()); m = d[b].max(0).values            # [32, 32] over channels
            ys, xs = torch.nonzero(m > 0.1 * m.max(), as_tuple=True)
            print(f"  copy {b}: {int((m > 0).sum())} of 1024 positions differ; rows {ys.min().item()}..{ys.max().item()}, cols {xs.min().item()}..{xs.max().item()} hold the large ones")
print(f"[{os.environ.get('MF_LIB_PATH', 'shipped library')}] batch {B}: {bad_calls} of {calls} calls with a copy or a call that differs; worst difference {worst:.3e}")
