#!/usr/bin/env python3
"""Builds the full-size MuseTalk step (BASELINE.json configs[2]: UNet + VAE decode, 256x256, batch 8) with seeded
weights, times it, and (optionally) checks one frame against the CPU oracle."""
import argparse, os, sys, time
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np, torch
from mere_fusion_amd import weights as W
from mere_fusion_amd.musetalk.models.unet import UNet
from mere_fusion_amd.musetalk.models.vae import VAE
from oracle import musetalk_ref as R

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--precision", default="bf16x3")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--check", type=int, default=1)
ap.add_argument("--small", type=int, default=0)
a = ap.parse_args()
cfg = R.MUSETALK_SMALL if a.small else R.MUSETALK_V1
t0 = time.time()
usd = W.make_musetalk_unet_state_dict(cfg, 0); vsd = W.make_musetalk_vae_state_dict(cfg, 0)
print(f"weights: unet {sum(v.numel() for v in usd.values()) / 1e6:.1f} M, vae {sum(v.numel() for v in vsd.values()) / 1e6:.1f} M params, {time.time() - t0:.1f} s", flush=True)
u = cfg["unet"]
ucfg = dict(in_channels=u["in_channels"], out_channels=u["out_channels"], block_out_channels=list(u["block_out_channels"]),
            layers_per_block=u["layers_per_block"], cross_attention_dim=u["cross_attention_dim"], attention_head_dim=u["attention_heads"],
            norm_num_groups=u["norm_num_groups"], down_attn=u["down_attn"], up_attn=u["up_attn"], sample_size=32)
t0 = time.time(); unet = UNet(ucfg, usd, precision=a.precision, max_batch=a.batch); print(f"unet create {time.time() - t0:.1f} s", flush=True)
vc = dict(cfg["vae"]); vc["block_out_channels"] = list(vc["block_out_channels"])
t0 = time.time(); vae = VAE(config=vc, state_dict=vsd, precision=a.precision, max_batch=a.batch); print(f"vae create {time.time() - t0:.1f} s", flush=True)
print(f"device memory in use: {torch.cuda.mem_get_info()[1] / 2**30 - torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB", flush=True)
lat, aud = W.make_musetalk_inputs(a.batch, 0)
latd, audd, t0d = lat.cuda(), aud.cuda(), torch.tensor([0]).cuda()

def step():
    pred = unet.model(latd, t0d, encoder_hidden_states=unet.pe(audd)).sample
    return pred, vae.decode_latents_device(pred)

for _ in range(3):
    pred, frames = step()
torch.cuda.synchronize()
for name, fn in (("unet", lambda: unet.model(latd, t0d, encoder_hidden_states=audd)), ("vae", lambda: vae.decode_latents_device(pred)), ("step", step)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.iters): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.iters
    print(f"{name}: {dt * 1e3:.2f} ms per batch of {a.batch} -> {a.batch / dt:.1f} frames/s", flush=True)
import ctypes as C
from mere_fusion_amd import _lib
l = _lib.lib()
if os.environ.get("MF_PROFILE_OPS"):
    for tag, h, nops, info, prof in (("unet", unet.model._h, l.mf_unet_num_ops, l.mf_unet_op_info, l.mf_unet_profile),
                                     ("vae", vae._h, l.mf_vae_num_ops, l.mf_vae_op_info, l.mf_vae_profile)):
        n = nops(h); ms = (C.c_float * n)()
        _lib.check(prof(h, a.batch, 3, ms, None))
        rows = []
        for i in range(n):
            nm, kn, fl = C.create_string_buffer(160), C.create_string_buffer(96), C.c_double()
            info(h, i, nm, 160, kn, 96, C.byref(fl)); rows.append((nm.value.decode(), kn.value.decode(), fl.value * a.batch, ms[i]))
        tot = sum(r[3] for r in rows)
        print(f"---- {tag}: {n} ops, sum {tot:.2f} ms")
        bykind = {}
        for nm, kn, fl, t in rows:
            k = bykind.setdefault(kn, [0.0, 0.0, 0]); k[0] += t; k[1] += fl; k[2] += 1
        for kn, (t, fl, cnt) in sorted(bykind.items(), key=lambda x: -x[1][0]):
            print(f"   {kn:44s} n={cnt:3d} {t:8.3f} ms {100 * t / tot:5.1f} %  {fl / (t * 1e-3) / 1e12 if t else 0:7.1f} TF")
        for nm, kn, fl, t in sorted(rows, key=lambda r: -r[3])[:12]:
            print(f"   top: {t * 1e3:8.1f} us {fl / 1e9:8.1f} GF {fl / (t * 1e-3) / 1e12:6.1f} TF  {kn:40s} {nm}")
m = R.count_macs(cfg)
print(f"algorithmic GFLOP/frame: unet {2 * m['unet'] / 1e9:.1f}, vae {2 * m['vae'] / 1e9:.1f}; at the step rate: {2 * (m['unet'] + m['vae']) * a.batch / dt / 1e12:.1f} TFLOP/s")
if a.check:
    torch.set_num_threads(min(16, os.cpu_count()))
    t0 = time.time()
    want_u8, want_pred = R.musetalk_step(usd, vsd, cfg, lat[:1], aud[:1])
    print(f"oracle one frame on CPU: {time.time() - t0:.1f} s")
    e = (pred[:1].cpu() - want_pred).abs().max().item()
    d = np.abs(frames[:1].cpu().numpy().astype(int) - want_u8.astype(int))
    print(f"parity frame 0: latents L-inf {e:.2e}; uint8 frame max diff {d.max()}, differing pixels {(d > 0).mean() * 100:.2f} %")
