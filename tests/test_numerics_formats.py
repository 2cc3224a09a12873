"""The operand formats behind `dtype` -- what ships (bf16x3) and what DESIGN.md proposes next (f16 + block-scaled FP8 / FP6 correction terms) --
simulated on the fp32 oracle of the MuseTalk step (tools/numerics_split_study.py; reduced-width config so the CPU tier stays fast; the full-size
table is in DESIGN.md).  The parity bound of BASELINE.json (fp32 L-inf <= 1e-3, uint8 frames within one level) must hold for every format the
product is allowed to compute in, and must FAIL for plain fp16 -- otherwise the bound would not be discriminating anything."""
import os
import sys

import pytest
import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def study():
    import numerics_split_study as S
    from mere_fusion_amd import weights as W
    from oracle import musetalk_ref as R
    cfg = R.MUSETALK_SMALL
    usd, vsd = W.make_musetalk_unet_state_dict(cfg, 0), W.make_musetalk_vae_state_dict(cfg, 0)
    lat, aud = W.make_musetalk_inputs(1, 0)
    conv0, lin0 = R._conv, R._lin
    out = {}
    try:
        for mode in ("fp32", "bf16x3", "f16+f8", "f16+f6", "f16x1"):
            R._conv, R._lin = (conv0, lin0) if mode == "fp32" else S.make_ops(mode)
            with torch.no_grad():
                pred = R.unet_forward(usd, cfg["unet"], lat, torch.tensor([0]), R.add_positional_encoding(aud))
                img = R.vae_decode(vsd, cfg["vae"], pred / cfg["vae"]["scaling_factor"])
            out[mode] = (pred, img, ((img / 2 + 0.5).clamp(0, 1) * 255).round())
    finally:
        R._conv, R._lin = conv0, lin0
    return out


@pytest.mark.parametrize("mode,passes", [("bf16x3", 3.0), ("f16+f8", 2.0), ("f16+f6", 1.5)])
def test_split_formats_stay_inside_the_parity_bound(study, mode, passes):
    p0, i0, u0 = study["fp32"]
    p, i, u = study[mode]
    lat_err, img_err = float((p - p0).abs().max()), float((i - i0).abs().max())
    d = (u - u0).abs()
    print(f"{mode} ({passes} MFMA pass-equivalents per product): latents {lat_err:.2e}, image {img_err:.2e}, uint8 max {int(d.max())}, {100 * float((d > 0).float().mean()):.2f} % pixels")
    assert lat_err <= 2.5e-4                     # a factor 4 inside the 1e-3 bound
    assert img_err / 2 <= 1e-3                   # the [-1, 1] image maps onto [0, 1] frames: halve
    assert d.max() <= 1 and float((d > 0).float().mean()) <= 0.01


def test_plain_fp16_breaks_the_bound(study):
    p0, i0, u0 = study["fp32"]
    p, i, u = study["f16x1"]
    assert float((p - p0).abs().max()) > 1e-3 and float(((u - u0).abs() > 0).float().mean()) > 0.05
