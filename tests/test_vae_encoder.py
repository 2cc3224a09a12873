"""Avatar preparation, encoder side: `VAE.preprocess_img` / `encode_latents` / `get_latents_for_unet`
(musetalk/models/vae.py:52-94, 110-122; called per avatar frame by mere_musetalk.py:303-304) on the GPU against oracle/musetalk_ref.py.
PARITY UNPINNED at the diffusers boundary; the checkpoint interface is pinned by the public parameter count."""
import numpy as np
import pytest
import torch

from mere_fusion_amd import weights as W
from mere_fusion_amd.musetalk.config import MUSETALK_V1, vae_config_json
from oracle import musetalk_ref as R


def test_encoder_manifest_completes_the_public_autoencoder_kl():
    enc = W.make_musetalk_vae_encoder_state_dict(MUSETALK_V1, 0, shapes_only=True)
    dec = W.make_musetalk_vae_state_dict(MUSETALK_V1, 0, shapes_only=True)
    n_enc = sum(v.numel() for k, v in enc.items() if k.startswith("encoder."))
    n_q = sum(v.numel() for k, v in enc.items() if k.startswith("quant_conv."))
    n_dec = sum(v.numel() for v in dec.values())
    assert (n_enc, n_q) == (34_163_592, 72)
    assert n_enc + n_q + n_dec == 83_653_863                        # sd-vae-ft-mse / SD-1.x AutoencoderKL, the published total
    assert tuple(enc["encoder.conv_out.weight"].shape) == (8, 512, 3, 3) and tuple(enc["quant_conv.weight"].shape) == (8, 8, 1, 1)
    assert tuple(enc["encoder.down_blocks.1.resnets.0.conv_shortcut.weight"].shape) == (256, 128, 1, 1)
    assert "encoder.down_blocks.3.downsamplers.0.conv.weight" not in enc and "encoder.down_blocks.2.downsamplers.0.conv.weight" in enc


def test_oracle_preprocess_kat():
    img = np.zeros((256, 256, 3), np.uint8)
    img[..., 0], img[..., 1], img[..., 2] = 255, 128, 0                # B, G, R
    x = R.preprocess_img(img, half_mask=True)
    assert x.shape == (1, 3, 256, 256)
    assert x[0, 0, 0, 0] == -1.0 and x[0, 2, 0, 0] == 1.0              # R = 0 -> -1, B = 255 -> +1: RGB order
    np.testing.assert_allclose(x[0, 1, 5, 5].item(), (128 / 255 - 0.5) / 0.5, rtol=0, atol=2e-7)   # fp32: 1 ulp of 0.5
    assert (x[0, :, 128:] == -1.0).all() and (x[0, 2, :128] == 1.0).all()   # masked half: zeros BEFORE the normalisation -> -1
    assert torch.equal(R.preprocess_img(img, half_mask=False)[0, 2], torch.ones(256, 256))


def test_oracle_downsample_pads_bottom_right_only():
    """Downsample2D of the encoder: output (i, j) reads input rows 2i .. 2i+2 -- the last one of the last row is the zero pad."""
    sd = {"d.weight": torch.zeros(1, 1, 3, 3), "d.bias": torch.zeros(1)}
    sd["d.weight"][0, 0, 0, 0] = 1.0                                   # picks input (2i, 2j)
    x = torch.arange(16.0).reshape(1, 1, 4, 4)
    y = R._conv(sd, "d", torch.nn.functional.pad(x, (0, 1, 0, 1)), stride=2, padding=0)
    assert y.shape == (1, 1, 2, 2) and y.flatten().tolist() == [0.0, 2.0, 8.0, 10.0]
    sd["d.weight"].zero_(); sd["d.weight"][0, 0, 2, 2] = 1.0           # picks (2i+2, 2j+2): pad for the last row / column
    y = R._conv(sd, "d", torch.nn.functional.pad(x, (0, 1, 0, 1)), stride=2, padding=0)
    assert y.flatten().tolist() == [10.0, 0.0, 0.0, 0.0]


def _vae(cfg, max_batch=2):
    from mere_fusion_amd.musetalk.models.vae import VAE
    sd = dict(W.make_musetalk_vae_state_dict(cfg, 0))
    sd.update(W.make_musetalk_vae_encoder_state_dict(cfg, 0))
    return VAE(config=vae_config_json(cfg["vae"]), state_dict=sd, max_batch=max_batch), sd


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["small", "v1"])
def test_hip_vae_encoder_vs_oracle(lib_built, name):
    cfg = R.MUSETALK_SMALL if name == "small" else MUSETALK_V1
    vae, sd = _vae(cfg)
    rng = np.random.default_rng(3)
    crops = rng.integers(0, 256, (2, 256, 256, 3), dtype=np.uint8)
    crops[1, 40:200, 60:180] = (crops[1, 40:200, 60:180] // 4 + 120)                      # some structure, not only noise
    torch.set_num_threads(16)
    for half in (False, True):
        want = torch.cat([R.vae_encode_moments(sd, cfg["vae"], R.preprocess_img(c, half_mask=half)) for c in crops])
        got_u8 = vae.encode_moments_device(image_u8_bgr=torch.from_numpy(crops).cuda(), half_mask=half).cpu()
        x = torch.cat([R.preprocess_img(c, half_mask=half) for c in crops]).cuda()
        got_f32 = vae.encode_moments_device(image=x).cpu()
        assert got_u8.shape == want.shape == (2, 8, 32, 32)
        scale = float(want.abs().max())
        e1, e2 = float((got_u8 - want).abs().max()), float((got_f32 - want).abs().max())
        print(f"{name} half_mask={half}: moments L-inf {e1:.3e} (u8 path) {e2:.3e} (fp32 path), |moments| max {scale:.2f}")
        assert e1 <= 1e-3 * max(1.0, scale) and e2 <= 1e-3 * max(1.0, scale)
        assert torch.equal(got_u8, got_f32)                                 # the device-side preprocessing is vae.py:52-82 bit for bit


@pytest.mark.gpu
def test_hip_get_latents_for_unet_with_the_same_noise(lib_built):
    """vae.py:110-122: [masked | reference] latents; `sample()` draws from torch's generator, so with the same draws the result is the oracle's."""
    cfg = R.MUSETALK_SMALL
    vae, sd = _vae(cfg)
    crop = np.random.default_rng(5).integers(0, 256, (256, 256, 3), dtype=np.uint8)
    g = torch.Generator(device="cuda").manual_seed(11)
    got = vae.get_latents_for_unet(crop, generator=g).cpu()
    g = torch.Generator(device="cuda").manual_seed(11)
    n1 = torch.randn((1, 4, 32, 32), generator=g, device="cuda").cpu()
    n2 = torch.randn((1, 4, 32, 32), generator=g, device="cuda").cpu()
    sf = cfg["vae"]["scaling_factor"]
    want = torch.cat([R.sample_latents(R.vae_encode_moments(sd, cfg["vae"], R.preprocess_img(crop, True)), sf, n1),
                      R.sample_latents(R.vae_encode_moments(sd, cfg["vae"], R.preprocess_img(crop, False)), sf, n2)], dim=1)
    assert got.shape == (1, 8, 32, 32)
    assert float((got - want).abs().max()) <= 2e-3 * max(1.0, float(want.abs().max()))
    x = vae.preprocess_img(crop, half_mask=True)
    assert x.is_cuda and torch.equal(x.cpu(), R.preprocess_img(crop, True))
    with pytest.raises(RuntimeError, match="stays with the reference"):
        vae.preprocess_img("some.png")
    from mere_fusion_amd.musetalk.models.vae import VAE
    dec_only = VAE(config=vae_config_json(cfg["vae"]), state_dict=W.make_musetalk_vae_state_dict(cfg, 0), max_batch=1)
    with pytest.raises(RuntimeError, match="decoder-only"):
        dec_only.encode_moments_device(image_u8_bgr=torch.zeros((1, 256, 256, 3), dtype=torch.uint8, device="cuda"))
