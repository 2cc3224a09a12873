"""Records tests/golden/musetalk_golden.npz from the REAL diffusers classes the reference instantiates -- on any box that has `diffusers` (this build
container does not: `pip install diffusers` is not possible here, so the fixture is absent and tests/test_oracle_golden.py::test_oracle_matches_diffusers_golden
skips; parity of rows a12 / a13 stays "unpinned" until someone runs this script once and commits the .npz).

    python tests/golden/make_musetalk_golden.py [--full]

What the reference does (musetalk/models/unet.py:29-44, musetalk/models/vae.py:17-36,96-108):
    UNet2DConditionModel(**json.load("musetalk.json"))          .forward(latents, timesteps, encoder_hidden_states).sample
    AutoencoderKL.from_pretrained("sd-vae-ft-mse")              .decode(latents / scaling_factor).sample -> (x / 2 + 0.5).clamp(0, 1) -> x 255 round -> uint8 BGR
Here both classes are built from the SAME config dicts and the SAME seeded state dicts the oracle and the HIP path use (mere_fusion_amd.weights.make_musetalk_*: diffusers
key names), at the reduced width of oracle.musetalk_ref.MUSETALK_SMALL (fixture of a few MB) -- with --full also at MUSETALK_V1 (outputs stored subsampled + as float64 sums).
Recorded (data only): the inputs' seeds, `latents_small` (1, 4, 32, 32), `image_small` (1, 3, 256, 256) pre-clamp, `u8_small`, and for --full strided samples and sums.
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

from mere_fusion_amd import weights as W                                   # noqa: E402
from oracle import musetalk_ref as R                                       # noqa: E402


def diffusers_unet_config(u):
    """oracle table -> the keyword arguments of diffusers.UNet2DConditionModel that musetalk.json holds (public MuseTalk v1 values for everything the table leaves out)"""
    down = tuple("CrossAttnDownBlock2D" if a else "DownBlock2D" for a in u["down_attn"])
    up = tuple("CrossAttnUpBlock2D" if a else "UpBlock2D" for a in u["up_attn"])
    return dict(sample_size=32, in_channels=u["in_channels"], out_channels=u["out_channels"], down_block_types=down, up_block_types=up,
                block_out_channels=tuple(u["block_out_channels"]), layers_per_block=u["layers_per_block"], cross_attention_dim=u["cross_attention_dim"],
                attention_head_dim=u["attention_heads"], norm_num_groups=u["norm_num_groups"], norm_eps=1e-5, act_fn="silu", flip_sin_to_cos=True, freq_shift=0,
                center_input_sample=False, downsample_padding=1, mid_block_scale_factor=1)


def diffusers_vae_config(v):
    n = len(v["block_out_channels"])
    return dict(in_channels=3, out_channels=v["out_channels"], down_block_types=("DownEncoderBlock2D",) * n, up_block_types=("UpDecoderBlock2D",) * n,
                block_out_channels=tuple(v["block_out_channels"]), layers_per_block=v["layers_per_block"], latent_channels=v["latent_channels"],
                norm_num_groups=v["norm_num_groups"], act_fn="silu", sample_size=256, scaling_factor=v["scaling_factor"])


def run(cfg, seed):
    from diffusers import UNet2DConditionModel, AutoencoderKL
    usd, vsd = W.make_musetalk_unet_state_dict(cfg, 0), W.make_musetalk_vae_state_dict(cfg, 0)
    unet = UNet2DConditionModel(**diffusers_unet_config(cfg["unet"])).eval()
    missing, unexpected = unet.load_state_dict(usd, strict=False)
    assert not missing and not unexpected, (missing[:5], unexpected[:5])
    vae = AutoencoderKL(**diffusers_vae_config(cfg["vae"])).eval()
    # the seeded VAE dict holds the decoder side (+ post_quant_conv); encoder tensors keep diffusers' own init and are not used
    missing, unexpected = vae.load_state_dict(vsd, strict=False)
    assert not unexpected and all(k.startswith(("encoder.", "quant_conv.")) for k in missing), (missing[:5], unexpected[:5])
    lat, aud = W.make_musetalk_inputs(1, seed)
    with torch.no_grad():
        pred = unet(lat, torch.tensor([0]), encoder_hidden_states=R.add_positional_encoding(aud)).sample
        img = vae.decode(pred / cfg["vae"]["scaling_factor"]).sample
    u8 = ((img / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1) * 255).round().to(torch.uint8).numpy()[..., ::-1].copy()      # vae.py:102-108 (RGB -> BGR)
    return pred.numpy(), img.numpy(), u8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    a = ap.parse_args()
    try:
        import diffusers
    except ImportError:
        raise SystemExit("diffusers is not installed here: run this script on a box that has it (the fixture is data, it travels)")
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    out = {"diffusers_version": np.array(diffusers.__version__), "input_seed": np.array(11)}
    pred, img, u8 = run(R.MUSETALK_SMALL, 11)
    out.update(latents_small=pred.astype(np.float32), image_small=img.astype(np.float32), u8_small=u8)
    if a.full:
        pred, img, u8 = run(R.MUSETALK_V1, 11)
        out.update(latents_full=pred.astype(np.float32), image_full_strided=img[:, :, ::7, ::5].astype(np.float32), image_full_sum=np.array(img.astype(np.float64).sum()),
                   u8_full_strided=u8[:, ::7, ::5])
    path = os.path.join(HERE, "musetalk_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
