"""CPU numerics study (build container or GPU box host, no GPU needed): how far can the MFMA passes per product be cut below bf16x3's three
while staying inside the parity bound?  Simulated on the fp32 oracle of the MuseTalk step (oracle/musetalk_ref.py) by replacing every
conv / linear product with the operand formats a CDNA4 kernel could issue:

  bf16x3      x = xh + xl (bf16 + bf16): wh*xh + wh*xl + wl*xh                                  3 bf16 passes   (what ships; control)
  f16x3       the same split with fp16 planes                                                   3 f16 passes
  f16+f8      xh, wh fp16; cross terms through FP8 e4m3: q8(wh)*q8(xl) + q8(wl)*q8(xh)          1 f16 + 2 fp8 passes; MX fp8 runs at 2x the bf16
                                                                                                 rate on gfx950 -> 2 pass-equivalents
  f16+f6      cross terms through FP6 e2m3 with per-32-element power-of-two block scales        1 f16 + 2 fp6 passes; fp6 runs at 4x -> 1.5
  f16x1       fp16 operands only (the reference's own .half() inference, musereal.py:60-62)     1 pass

Accumulation is fp32 everywhere (torch CPU conv).  Reported: UNet latent L-inf, VAE pre-clamp image L-inf, uint8 frame differences vs the fp32
oracle, one frame.    python tools/numerics_split_study.py [--small]"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mere_fusion_amd import weights as W            # noqa: E402
from oracle import musetalk_ref as R                 # noqa: E402


def q_bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def q_f16(t):
    return t.to(torch.float16).to(torch.float32)


def q_f8(t):
    """e4m3 with one power-of-two scale per tensor (the residual planes span few binades)"""
    m = t.abs().max().clamp_min(1e-30)
    s = 2.0 ** torch.floor(torch.log2(256.0 / m))
    return (t * s).to(torch.float8_e4m3fn).to(torch.float32) / s


def q_f6(t, axis):
    """e2m3 (values +-{0, .125 ... 7.5}) with an E8M0 scale per block of 32 along the contraction axis (OCP MX)"""
    t = t.movedim(axis, -1)
    shp = t.shape
    pad = (-shp[-1]) % 32
    x = F.pad(t, (0, pad)).reshape(*shp[:-1], -1, 32)
    m = x.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    s = 2.0 ** (torch.floor(torch.log2(m)) - 2)                         # block max lands in [4, 8)
    y = x / s
    mag = y.abs()
    e = torch.floor(torch.log2(mag.clamp_min(1e-30))).clamp(0, 2)      # normal binades 1, 2, 4; below 1: subnormal step 0.125
    step = torch.where(mag < 1.0, torch.full_like(mag, 0.125), 2.0 ** e / 8)
    q = (torch.round(mag / step) * step).clamp(max=7.5) * torch.sign(y)
    out = (q * s).reshape(*shp[:-1], -1)[..., : shp[-1]]
    return out.movedim(-1, axis)


def make_ops(mode):
    def product(op, x, w, caxis_x, caxis_w):
        if mode == "fp32":
            return op(x, w)
        if mode == "f16x1":
            return op(q_f16(x), q_f16(w))
        if mode == "xf16+w2":                            # x rounded to f16 (one plane), w = f16 + f16: two f16 passes, 2 + 4 operand bytes
            xh, wh = q_f16(x), q_f16(w)
            return op(xh, wh) + op(xh, q_f16(w - wh))
        hi = q_bf16 if mode == "bf16x3" else q_f16
        xh, wh = hi(x), hi(w)
        xl, wl = x - xh, w - wh
        if mode in ("bf16x3", "f16x3"):
            return op(xh, wh) + op(hi(xl), wh) + op(xh, hi(wl))
        if mode == "f16+f8":
            return op(xh, wh) + op(q_f8(xl), q_f8(wh)) + op(q_f8(xh), q_f8(wl))
        if mode == "f16+f6":
            return op(xh, wh) + op(q_f6(xl, caxis_x), q_f6(wh, caxis_w)) + op(q_f6(xh, caxis_x), q_f6(wl, caxis_w))
        raise ValueError(mode)

    def conv(sd, p, x, stride=1, padding=1):
        y = product(lambda a, b: F.conv2d(a, b, None, stride=stride, padding=padding), x, sd[p + ".weight"], 1, 1)
        b = sd.get(p + ".bias")
        return y if b is None else y + b.view(1, -1, 1, 1)

    def lin(sd, p, x):
        y = product(lambda a, b: F.linear(a, b), x, sd[p + ".weight"], x.dim() - 1, 1)
        b = sd.get(p + ".bias")
        return y if b is None else y + b
    return conv, lin


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--small", action="store_true", help="reduced-width config (seconds instead of minutes)")
    ap.add_argument("--modes", default="", help="comma list; U/V = UNet format / VAE format, e.g. fp32,bf16x3/f16+f6,xf16+w2/f16+f6")
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = R.MUSETALK_SMALL if a.small else R.MUSETALK_V1
    usd, vsd = W.make_musetalk_unet_state_dict(cfg, 0), W.make_musetalk_vae_state_dict(cfg, 0)
    lat, aud = W.make_musetalk_inputs(1, 0)
    conv0, lin0 = R._conv, R._lin
    res = {}
    modes = a.modes.split(",") if a.modes else ["fp32", "bf16x3", "f16x3", "f16+f8", "f16+f6", "f16x1"]
    for mode in modes:
        # "U/V": the UNet in format U, the VAE decoder in format V (what ships is bf16x3/f16+f6)
        mu, mv = mode.split("/") if "/" in mode else (mode, mode)
        with torch.no_grad():
            R._conv, R._lin = (conv0, lin0) if mu == "fp32" else make_ops(mu)
            pred = R.unet_forward(usd, cfg["unet"], lat, torch.tensor([0]), R.add_positional_encoding(aud))
            R._conv, R._lin = (conv0, lin0) if mv == "fp32" else make_ops(mv)
            img = R.vae_decode(vsd, cfg["vae"], pred / cfg["vae"]["scaling_factor"])
        u8 = ((img / 2 + 0.5).clamp(0, 1) * 255).round()
        res[mode] = (pred, img, u8)
        if mode != "fp32":
            p0, i0, u0 = res["fp32"]
            d = (u8 - u0).abs()
            print(f"{mode:8s} latents L-inf {float((pred - p0).abs().max()):.2e}   image (pre-clamp, [-1, 1]) L-inf {float((img - i0).abs().max()):.2e}   "
                  f"uint8: max diff {int(d.max())}, differing pixels {100 * float((d > 0).float().mean()):.3f} %", flush=True)
    R._conv, R._lin = conv0, lin0


if __name__ == "__main__":
    main()
