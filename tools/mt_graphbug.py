import os, sys
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np, torch
from mere_fusion_amd import weights as W
from mere_fusion_amd.musetalk.models.unet import UNet
from mere_fusion_amd.musetalk.models.vae import VAE
from oracle import musetalk_ref as R
cfg = R.MUSETALK_SMALL
vsd = W.make_musetalk_vae_state_dict(cfg, 0)
vc = dict(cfg["vae"]); vc["block_out_channels"] = list(vc["block_out_channels"]); vae = VAE(config=vc, state_dict=vsd, max_batch=4)
torch.manual_seed(0); A = torch.randn(2, 4, 32, 32) * 0.2; Bt = torch.randn(2, 4, 32, 32) * 0.3
wA = R.decode_latents(vsd, cfg["vae"], A); wB = R.decode_latents(vsd, cfg["vae"], Bt)
usd = W.make_musetalk_unet_state_dict(cfg, 0); u = cfg["unet"]
ucfg = dict(in_channels=8, out_channels=4, block_out_channels=list(u["block_out_channels"]), layers_per_block=2, cross_attention_dim=384,
            attention_head_dim=8, norm_num_groups=32, down_attn=u["down_attn"], up_attn=u["up_attn"], sample_size=32)
unet = UNet(ucfg, usd, max_batch=4) if os.environ.get("WITH_UNET") else None
lat, aud = W.make_musetalk_inputs(2, 7)
seq = os.environ.get("SEQ", "AABB")
if os.environ.get("IMG"):
    fr, im = vae.decode_latents_device(A.cuda(), want_image=True); print("I max diff", int(np.abs(fr.cpu().numpy().astype(int) - wA.astype(int)).max()))
    seq = seq[1:]
for ch in seq:
    if ch == "V":
        global_v2 = globals().setdefault("vae2", None) or VAE(config=vc, state_dict=vsd, max_batch=4); globals()["vae2"] = global_v2
        o2 = global_v2.decode_latents(A.cuda()); print("V max diff", int(np.abs(o2.astype(int) - wA.astype(int)).max())); continue
    if ch == "S":
        torch.cuda.synchronize(); x = torch.randn(64, 1024, 1024, device="cuda"); y = (x @ x).sum().item(); print("S torch work"); continue
    if ch == "U":
        pred = unet.model(lat.cuda(), torch.tensor([0]).cuda(), encoder_hidden_states=aud.cuda()).sample; print("U", float(pred.abs().max())); continue
    x, w = (A, wA) if ch == "A" else (Bt, wB)
    o = vae.decode_latents(x.cuda())
    print(ch, "max diff vs oracle", int(np.abs(o.astype(int) - w.astype(int)).max()), flush=True)
