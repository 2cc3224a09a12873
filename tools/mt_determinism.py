import os, sys
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np, torch
from mere_fusion_amd import weights as W
from mere_fusion_amd.musetalk.models.unet import UNet
from mere_fusion_amd.musetalk.models.vae import VAE
from oracle import musetalk_ref as R
cfg = R.MUSETALK_SMALL
usd = W.make_musetalk_unet_state_dict(cfg, 0); vsd = W.make_musetalk_vae_state_dict(cfg, 0)
u = cfg["unet"]
ucfg = dict(in_channels=8, out_channels=4, block_out_channels=list(u["block_out_channels"]), layers_per_block=2, cross_attention_dim=384,
            attention_head_dim=8, norm_num_groups=32, down_attn=u["down_attn"], up_attn=u["up_attn"], sample_size=32)
unet = UNet(ucfg, usd, max_batch=4); vc = dict(cfg["vae"]); vc["block_out_channels"] = list(vc["block_out_channels"]); vae = VAE(config=vc, state_dict=vsd, max_batch=4)
lat, aud = W.make_musetalk_inputs(2, 7)
want, wp = R.musetalk_step(usd, vsd, cfg, lat, aud)
preds, outs = [], []
for i in range(5):
    pred = unet.model(lat.cuda(), torch.tensor([0]).cuda(), encoder_hidden_states=aud.cuda()).sample
    preds.append(pred.cpu()); outs.append(vae.decode_latents(pred))
for i in range(5):
    print(i, "pred vs oracle", float((preds[i] - wp).abs().max()), "pred vs run0", float((preds[i] - preds[0]).abs().max()),
          "u8 vs oracle max", int(np.abs(outs[i].astype(int) - want.astype(int)).max()), "u8 vs run0 max", int(np.abs(outs[i].astype(int) - outs[0].astype(int)).max()))
# vae alone, same latents
fr = [vae.decode_latents(preds[0].cuda()) for _ in range(4)]
print("vae alone vs first:", [int(np.abs(f.astype(int) - fr[0].astype(int)).max()) for f in fr], "vs oracle", int(np.abs(fr[-1].astype(int) - R.decode_latents(vsd, cfg["vae"], preds[0]).astype(int)).max()))
print("---- scenario 2: VAE graph captured on other latents first")
unet2 = UNet(ucfg, usd, max_batch=4); vae2 = VAE(config=vc, state_dict=vsd, max_batch=4)
torch.manual_seed(0); l0 = (torch.randn(2, 4, 32, 32) * 0.2).cuda()
vae2.decode_latents_device(l0, want_image=True); vae2.decode_latents(l0)
preds, outs = [], []
for i in range(4):
    pred = unet2.model(lat.cuda(), torch.tensor([0]).cuda(), encoder_hidden_states=aud.cuda()).sample
    preds.append(pred.cpu()); outs.append(vae2.decode_latents(pred))
for i in range(4):
    print(i, "pred vs run0", float((preds[i] - preds[0]).abs().max()), "pred vs oracle", float((preds[i] - wp).abs().max()),
          "u8 vs run0", int(np.abs(outs[i].astype(int) - outs[0].astype(int)).max()), "u8 vs oracle", int(np.abs(outs[i].astype(int) - want.astype(int)).max()))
fr = [vae2.decode_latents(preds[0].cuda()) for _ in range(3)]
print("vae2 replays on fixed latents vs first:", [int(np.abs(f.astype(int) - fr[0].astype(int)).max()) for f in fr], "vs oracle", int(np.abs(fr[-1].astype(int) - want.astype(int)).max()))
