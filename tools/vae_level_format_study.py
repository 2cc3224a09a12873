#!/usr/bin/env python3
"""CPU numerics study for VERDICT r05 item 4: the VAE phase is energy-bound, so the lever left is FEWER matrix instructions per output -- can some
LEVEL of the decoder run a cheaper operand format than f16 + FP6 (1.5 pass-equivalents) and stay inside the parity gates?  Candidates, per level of the decoder
(mid / up0 at 32^2 ... up3 at 256^2: the six 404 us launches of the roofline kernel):

  f16x1      x, w rounded to fp16, nothing else                       1 pass
  f16+wl     xh.wh + q6(xh).q6(wl)   (the WEIGHT residual only)       1.25
  f16+xl     xh.wh + q6(xl).q6(wh)   (the ACTIVATION residual only)   1.25
  f16+f6     both residuals (what ships)                              1.5

Simulated on the fp32 oracle (oracle/musetalk_ref.py) as tools/numerics_split_study.py does, one format per layer chosen by the layer's name.  Two weight sets:
the seeded sd-vae-ft-mse decoder, and the same decoder under tests/test_musetalk_stress.py's re-parametrisation (per-channel scales over four decades + 1 % outliers
x 30).  Under stress the shipped format relies on the load-time channel equalisation of its MX blocks, which this simulation does not model; there the levels that
keep the shipped format are run as bf16x3 (insensitive to the stress: 7e-5), so the number is the MARGINAL error of the candidate level.
Adoption rule (VERDICT): image L-inf <= 5e-4 and uint8 max diff 1 on <= 0.7 % of the pixels, in both studies.

    python tools/vae_level_format_study.py [--small] [--frames 2]            (minutes on 8 cores at full size)"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from mere_fusion_amd import weights as W            # noqa: E402
from oracle import musetalk_ref as R                 # noqa: E402
from numerics_split_study import q_bf16, q_f16, q_f6  # noqa: E402

LEVELS = {"mid": ("decoder.mid_block.", "decoder.conv_in"), "up0": ("decoder.up_blocks.0.",), "up1": ("decoder.up_blocks.1.",), "up2": ("decoder.up_blocks.2.",),
          "up3": ("decoder.up_blocks.3.",)}


def product(mode, op, x, w):
    if mode == "fp32":
        return op(x, w)
    if mode == "bf16x3":
        xh, wh = q_bf16(x), q_bf16(w)
        return op(xh, wh) + op(q_bf16(x - xh), wh) + op(xh, q_bf16(w - wh))
    xh, wh = q_f16(x), q_f16(w)
    y = op(xh, wh)
    if mode in ("f16+f6", "f16+xl"):
        y = y + op(q_f6(x - xh, 1), q_f6(wh, 1))
    if mode in ("f16+f6", "f16+wl"):
        y = y + op(q_f6(xh, 1), q_f6(w - wh, 1))
    return y


def make_conv(plan, default):
    """plan: {level name: mode}; layers of other levels (and 1x1 / attention / conv_out layers: they are bf16x3 in the product) run `default` / bf16x3"""
    def mode_of(p, w):
        if w.shape[-1] != 3 or p == "decoder.conv_out":
            return "bf16x3"                                          # shortcuts, attention projections, post_quant_conv, the fused tail: bf16x3 kernels in the product
        for lv, prefixes in LEVELS.items():
            if any(p.startswith(q) for q in prefixes):
                return plan.get(lv, default)
        return default

    def conv(sd, p, x, stride=1, padding=1):
        w = sd[p + ".weight"]
        y = product(mode_of(p, w), lambda a, b: F.conv2d(a, b, None, stride=stride, padding=padding), x, w)
        b = sd.get(p + ".bias")
        return y if b is None else y + b.view(1, -1, 1, 1)

    def lin(sd, p, x):
        y = product("bf16x3", lambda a, b: F.linear(a, b), x, sd[p + ".weight"])
        b = sd.get(p + ".bias")
        return y if b is None else y + b
    return conv, lin


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--frames", type=int, default=2)
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = R.MUSETALK_SMALL if a.small else R.MUSETALK_V1
    vsd0 = W.make_musetalk_vae_state_dict(cfg, 0)
    from test_musetalk_stress import stressed_vae_state_dict
    vsd_s, _ = stressed_vae_state_dict(vsd0, 1)
    lat = torch.randn(a.frames, 4, 32, 32, generator=torch.Generator().manual_seed(41)) * 0.18215 / cfg["vae"]["scaling_factor"]
    conv0, lin0 = R._conv, R._lin
    plans = [("f16+f6 everywhere (ships)", {}, "f16+f6"),
             ("up3 (256^2) f16x1", {"up3": "f16x1"}, "f16+f6"),
             ("up3 f16+wl", {"up3": "f16+wl"}, "f16+f6"),
             ("up3 f16+xl", {"up3": "f16+xl"}, "f16+f6"),
             ("up2 + up3 f16+wl", {"up2": "f16+wl", "up3": "f16+wl"}, "f16+f6"),
             ("up2 + up3 f16+xl", {"up2": "f16+xl", "up3": "f16+xl"}, "f16+f6"),
             ("every level f16+wl", {k: "f16+wl" for k in LEVELS}, "f16+f6"),
             ("every level f16+xl", {k: "f16+xl" for k in LEVELS}, "f16+f6"),
             ("every level f16x1", {k: "f16x1" for k in LEVELS}, "f16+f6")]
    passes = {"f16x1": 1.0, "f16+wl": 1.25, "f16+xl": 1.25, "f16+f6": 1.5}
    # algorithmic GFLOP per frame-batch of 8 of the f16 + FP6 launches per level (profiles/r05_layers.json: mid 4 x 38.6, up0 6 x 38.6, up1 928 + its upsampler 618,
    # up2 1082 + 618, up3 1082): the "pass-equivalents" column weighs the levels by it
    work = {"mid": 154.6, "up0": 232.0, "up1": 1546.0, "up2": 1700.0, "up3": 1082.0}
    total = sum(work.values())
    print(f"VAE decoder, {'reduced' if a.small else 'sd-vae-ft-mse'} config, {a.frames} frame(s); gates: image L-inf <= 5e-4, uint8 max diff 1 on <= 0.7 %\n")
    print("| plan | MFMA pass-equivalents of the 3x3 convs (ships = 1.5) | seeded weights: image L-inf | uint8 max / % differing | stressed weights (marginal, other levels bf16x3): image L-inf | uint8 max / % | verdict |")
    print("|---|---:|---:|---:|---:|---:|---|")
    ref = {}
    for tag, sd in (("seed", vsd0), ("stress", vsd_s)):
        with torch.no_grad():
            R._conv, R._lin = conv0, lin0
            img = R.vae_decode(sd, cfg["vae"], lat)
        ref[tag] = (img, ((img / 2 + 0.5).clamp(0, 1) * 255).round())
    for name, plan, default in plans:
        t0 = time.time()
        cells = []
        ok = True
        for tag, sd in (("seed", vsd0), ("stress", vsd_s)):
            dflt = default if tag == "seed" else "bf16x3"
            with torch.no_grad():
                R._conv, R._lin = make_conv(plan, dflt)
                img = R.vae_decode(sd, cfg["vae"], lat)
            u8 = ((img / 2 + 0.5).clamp(0, 1) * 255).round()
            e = float((img - ref[tag][0]).abs().max())
            d = (u8 - ref[tag][1]).abs()
            frac = 100 * float((d > 0).float().mean())
            ok = ok and e <= 5e-4 and int(d.max()) <= 1 and frac <= 0.7
            cells += [f"{e:.2e}", f"{int(d.max())} / {frac:.3f}"]
        pe = sum(work[lv] * passes[plan.get(lv, "f16+f6")] for lv in LEVELS) / total
        print(f"| {name} | {pe:.3f} | {cells[0]} | {cells[1]} | {cells[2]} | {cells[3]} | {'inside the gates' if ok else 'REJECTED'} |   <!-- {time.time() - t0:.0f} s -->", flush=True)
    R._conv, R._lin = conv0, lin0


if __name__ == "__main__":
    main()
