"""Dynamic-range stress of the VAE decoder's f16 + FP6 (MX-block) operand format (VERDICT r03 missing #4 / weak #2).

Every other parity number in this suite is on seeded random-init weights: activations of one scale in every channel.  The f16 + FP6 format quantises the RESIDUAL of
each value in blocks of 32 channels sharing one E8M0 scale, after GroupNorm + SiLU -- a trained decoder has channels of very different magnitude and a few outlier
channels, and a block's scale follows its largest member.  Here the seeded sd-vae-ft-mse decoder is re-scaled so that it has them: for every (GroupNorm -> SiLU ->
conv) pair, channel c of the norm's affine (gamma, beta) is multiplied by s_c and input channel c of the conv by 1 / s_c, with s_c log-uniform in [1e-2, 1e2] and
1 % of the channels another x 30 -- the small-activation channels carry LARGE weights, so what the block scale rounds away in them matters as much as what it keeps
in the outliers.  Same gates as tests/test_musetalk_full.py, against the fp32 oracle on the same re-scaled weights."""
import numpy as np
import pytest
import torch

from mere_fusion_amd import weights as W
from mere_fusion_amd.musetalk.config import MUSETALK_V1, vae_config_json

pytestmark = pytest.mark.gpu

TOL_IMAGE = 6e-4           # the gate of tests/test_musetalk_full.py (bound: 1e-3 relative to values in about [-4, 4])
TOL_U8_FRACTION = 0.007


def stressed_vae_state_dict(vsd, seed, lo=1e-2, hi=1e2, outlier_frac=0.01, outlier_gain=30.0, one_sided=False):
    """one_sided: only the conv's input channels are re-scaled (by s_c / geometric mean) -- the network changes, nothing in front of the conv compensates: channels of
    unequal IMPORTANCE rather than a re-parametrisation."""
    g = torch.Generator().manual_seed(seed)
    sd = {k: v.clone() for k, v in vsd.items()}
    pairs = [(k[:-len(".norm1.weight")] + ".norm1", k[:-len(".norm1.weight")] + ".conv1") for k in sd if k.startswith("decoder.") and k.endswith(".norm1.weight")]
    pairs += [(k[:-len(".norm2.weight")] + ".norm2", k[:-len(".norm2.weight")] + ".conv2") for k in sd if k.startswith("decoder.") and k.endswith(".norm2.weight")]
    pairs.append(("decoder.conv_norm_out", "decoder.conv_out"))
    for norm, conv in sorted(pairs):
        c = sd[norm + ".weight"].numel()
        s = torch.exp(torch.rand(c, generator=g) * (np.log(hi) - np.log(lo)) + np.log(lo))
        s[torch.rand(c, generator=g) < outlier_frac] *= outlier_gain
        if one_sided:
            s = s / torch.exp(torch.log(s).mean())
            sd[conv + ".weight"] = sd[conv + ".weight"] * (s / s.pow(2).mean().sqrt())[None, :, None, None]      # (same overall gain)
            continue
        sd[norm + ".weight"] = sd[norm + ".weight"] * s
        sd[norm + ".bias"] = sd[norm + ".bias"] * s
        sd[conv + ".weight"] = sd[conv + ".weight"] / s[None, :, None, None]
    return sd, len(pairs)


@pytest.mark.parametrize("seed,one_sided", [(1, False), (2, False), (1, True)])
def test_vae_f16_fp6_under_channel_scale_stress(lib_built, seed, one_sided):
    """Measured on MI355X (tools/vae_stress_probe.py), image L-inf vs the oracle, scales over [1e-2, 1e2] + 1 % outliers x 30:
         re-parametrised   f16 + FP6 with the load-time channel equalisation (shipped)  1.2e-4      without it (MF_Q_EQUALIZE=0)  1.5e-3, 2.5 % of uint8 pixels off by one
                           bf16x3 everywhere (MF_CONV_Q=0)                               7e-5
         one-sided         1.7e-4 / 1.9e-4 / 1.4e-4
       (no stress: 1.4e-4 / 1.4e-4 / 9e-5).  The equalisation exists because of this test."""
    import os
    from mere_fusion_amd.musetalk.models.vae import VAE
    from oracle import musetalk_ref as R
    B = 8                                                      # (the batch-8 handle: every resnet conv of the 64^2 ... 256^2 levels on the f16 + FP6 halo tile)
    vsd0 = W.make_musetalk_vae_state_dict(MUSETALK_V1, 0)
    vsd, n_pairs = stressed_vae_state_dict(vsd0, seed, one_sided=one_sided)
    assert n_pairs >= 25
    vae = VAE(config=vae_config_json(MUSETALK_V1["vae"]), state_dict=vsd, max_batch=B)
    lat = (torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(40 + seed)) * 0.18215).repeat(B // 2, 1, 1, 1)
    torch.set_num_threads(max(1, min(len(os.sched_getaffinity(0)), 32)))
    want_img = R.vae_decode(vsd, MUSETALK_V1["vae"], lat[:2] / MUSETALK_V1["vae"]["scaling_factor"])
    want_u8 = R.decode_latents(vsd, MUSETALK_V1["vae"], lat[:2])
    frames, image = vae.decode_latents_device(lat.cuda(), want_image=True)
    scale = float(want_img.abs().max())
    ierr = (image.cpu()[:2] - want_img).abs().max().item()
    d = np.abs(frames.cpu().numpy()[:2].astype(int) - want_u8.astype(int))
    print(f"stressed sd-vae-ft-mse decoder ({'one-sided' if one_sided else 're-parametrised'}, seed {seed}, {n_pairs} re-scaled norm -> conv pairs): image L-inf {ierr:.3e} "
          f"on values up to {scale:.2f} (gate {TOL_IMAGE}); uint8 max diff {d.max()}, differing pixels {100 * (d > 0).mean():.3f} %")
    assert np.isfinite(ierr) and ierr <= TOL_IMAGE, (ierr, scale)
    assert d.max() <= 1 and (d > 0).mean() < TOL_U8_FRACTION, (d.max(), (d > 0).mean())


def test_vae_on_a_map_that_is_not_a_multiple_of_64_pixels(lib_built):
    """maps whose pixel count is not a multiple of 64 take the per-thread-parameter conversion kernel (k_affine_silu_to_q, the fallback of
    mf_affine_silu_to_act_q) and ragged 16 x 16 tiles of the f16 + FP6 conv; nothing at the BASELINE sizes does.  A 9 x 9 latent grid (18^2, 36^2, 72^2 maps:
    36^2 = 1296 = 20.25 x 64 pixels, 512 channels, two and a quarter tiles per side) against the fp32 oracle, same gates as the full-size decoder."""
    from mere_fusion_amd.musetalk.models.vae import VAE
    from mere_fusion_amd import _lib
    from oracle import musetalk_ref as R
    import ctypes as C
    B = 4
    vsd = W.make_musetalk_vae_state_dict(MUSETALK_V1, 0)
    cfg = vae_config_json(MUSETALK_V1["vae"])
    cfg["latent_size"] = 9
    vae = VAE(config=cfg, state_dict=vsd, max_batch=B)
    # the op list of the handle must show the conversion pass on the 36 x 36 level (otherwise this test no longer covers the fallback kernel)
    l = _lib.lib()
    n = l.mf_vae_num_ops(vae._h)
    kernels = []
    for i in range(n):
        nm, kn, fl = C.create_string_buffer(160), C.create_string_buffer(160), C.c_double()
        l.mf_vae_op_info(vae._h, i, nm, 160, kn, 160, C.byref(fl))
        kernels.append(kn.value.decode())
    assert any("k_affine_silu_to_q" in k for k in kernels), kernels
    lat = (torch.randn(2, 4, 9, 9, generator=torch.Generator().manual_seed(77)) * 0.18215).repeat(B // 2, 1, 1, 1)
    want_img = R.vae_decode(vsd, MUSETALK_V1["vae"], lat[:2] / MUSETALK_V1["vae"]["scaling_factor"])
    want_u8 = R.decode_latents(vsd, MUSETALK_V1["vae"], lat[:2])
    frames, image = vae.decode_latents_device(lat.cuda(), want_image=True)
    assert tuple(frames.shape) == (B, 72, 72, 3)
    ierr = (image.cpu()[:2] - want_img).abs().max().item()
    d = np.abs(frames.cpu().numpy()[:2].astype(int) - want_u8.astype(int))
    same = (frames[:2] == frames[2:]).all().item()
    print(f"sd-vae-ft-mse decoder on a 9 x 9 latent grid ({n} ops): image L-inf {ierr:.3e} (gate {TOL_IMAGE}); uint8 max diff {d.max()}, differing pixels {100 * (d > 0).mean():.3f} %; "
          f"copies bit-identical: {same}")
    assert np.isfinite(ierr) and ierr <= TOL_IMAGE
    assert d.max() <= 1 and (d > 0).mean() < TOL_U8_FRACTION
    assert same


def test_groupnorm_affine_inside_the_conversion_kernel_is_bit_identical(lib_built, monkeypatch):
    """The VAE decoder's GroupNorm -> SiLU -> f16 + FP6 conversions form the per-channel affine inside the conversion kernel (mf_gn_affine_pair, one channel per
    lane, handed to the scalar registers by v_readlane) instead of a k_gn_affine launch in front of each: the same frames and the same fp32 image, bit for bit,
    as the two-launch form (MF_GN_AFFINE_FUSE=0) -- on a reduced 16 x 16 latent grid (maps 32^2 ... 128^2: every level a multiple of 64 pixels) and with the
    four-decade stress weights."""
    from mere_fusion_amd.musetalk.models.vae import VAE
    from mere_fusion_amd import _lib
    import ctypes as C
    B = 2
    vsd, _ = stressed_vae_state_dict(W.make_musetalk_vae_state_dict(MUSETALK_V1, 0), 3)
    cfg = vae_config_json(MUSETALK_V1["vae"])
    cfg["latent_size"] = 16
    lat = (torch.randn(B, 4, 16, 16, generator=torch.Generator().manual_seed(5)) * 0.18215).cuda()
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MF_GN_AFFINE_FUSE", mode)
        vae = VAE(config=cfg, state_dict=vsd, max_batch=B)
        l = _lib.lib()
        names = []
        for i in range(l.mf_vae_num_ops(vae._h)):
            nm, kn, fl = C.create_string_buffer(160), C.create_string_buffer(160), C.c_double()
            l.mf_vae_op_info(vae._h, i, nm, 160, kn, 160, C.byref(fl))
            names.append(kn.value.decode())
        assert any("k_affine_silu_to_q" in k for k in names), names
        frames, image = vae.decode_latents_device(lat, want_image=True)
        torch.cuda.synchronize()
        outs[mode] = (frames.clone(), image.clone())
        del vae
    assert torch.equal(outs["1"][0], outs["0"][0]) and torch.equal(outs["1"][1], outs["0"][1])
    assert outs["1"][1].float().std().item() > 0.01
