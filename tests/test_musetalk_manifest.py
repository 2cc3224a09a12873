"""Checkpoint manifest of the MuseTalk stage: the state-dict key names / shapes the HIP loaders (csrc/mf_musetalk.hip) and the oracle
(oracle/musetalk_ref.py) consume, held against facts about the PUBLIC checkpoints that are independent of this repository:

  * Stable-Diffusion-1.x `UNet2DConditionModel` (diffusers): 686 state-dict tensors, 859,520,964 parameters.  MuseTalk's UNet is that
    network with `in_channels` 8 instead of 4 (conv_in gains 4 * 320 * 9 = 11,520 weights) and `cross_attention_dim` 384 instead of 768
    (each of the 16 cross-attention blocks loses 2 * C * 384 weights in to_k / to_v: 5 blocks at C = 320, 5 at 640, 6 at 1280
    = 9,584,640): 859,520,964 + 11,520 - 9,584,640 = 849,947,844.
  * sd-vae-ft-mse `AutoencoderKL`: 83,653,863 parameters, of which the decoder holds 49,490,179 (+ 20 in post_quant_conv).
  * A list of individual (name, shape) pairs every SD-1.x checkpoint contains, written from the published checkpoint layout.
PARITY still UNPINNED at the diffusers boundary (no diffusers, no config JSON here): this pins the checkpoint INTERFACE -- a real
`pytorch_model.bin` / `diffusion_pytorch_model.bin` provably maps onto the loaders -- not the arithmetic."""
import re

import pytest
import torch

from mere_fusion_amd import weights as W
from mere_fusion_amd.musetalk.config import MUSETALK_V1

SD1X_UNET_PARAMS, SD1X_UNET_TENSORS = 859_520_964, 686
SD_VAE_DECODER_PARAMS, SD_VAE_POST_QUANT_PARAMS = 49_490_179, 20


@pytest.fixture(scope="module")
def unet_manifest():
    return {k: tuple(v.shape) for k, v in W.make_musetalk_unet_state_dict(MUSETALK_V1, 0, shapes_only=True).items()}


@pytest.fixture(scope="module")
def vae_manifest():
    return {k: tuple(v.shape) for k, v in W.make_musetalk_vae_state_dict(MUSETALK_V1, 0, shapes_only=True).items()}


def _numel(shapes):
    n = 0
    for s in shapes.values():
        m = 1
        for d in s:
            m *= d
        n += m
    return n


def test_unet_tensor_and_parameter_counts_match_public_sd1x(unet_manifest):
    assert len(unet_manifest) == SD1X_UNET_TENSORS
    cross_blocks = {320: 5, 640: 5, 1280: 6}                                   # down 2 + up 3, down 2 + up 3, down 2 + mid 1 + up 3
    delta = 4 * 320 * 9 - sum(2 * c * (768 - 384) * n for c, n in cross_blocks.items())
    assert delta == 11_520 - 9_584_640
    assert _numel(unet_manifest) == SD1X_UNET_PARAMS + delta == 849_947_844


def test_unet_contains_the_published_sd1x_entries(unet_manifest):
    known = {
        "conv_in.weight": (320, 8, 3, 3), "conv_in.bias": (320,),
        "time_embedding.linear_1.weight": (1280, 320), "time_embedding.linear_2.weight": (1280, 1280),
        "down_blocks.0.resnets.0.norm1.weight": (320,), "down_blocks.0.resnets.0.conv1.weight": (320, 320, 3, 3),
        "down_blocks.0.resnets.0.time_emb_proj.weight": (320, 1280),
        "down_blocks.0.attentions.0.norm.weight": (320,), "down_blocks.0.attentions.0.proj_in.weight": (320, 320, 1, 1),     # conv projections (SD 1.x)
        "down_blocks.0.attentions.0.transformer_blocks.0.norm1.weight": (320,),
        "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.weight": (320, 320),
        "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_out.0.bias": (320,),
        "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_k.weight": (320, 384),                                    # 768 in SD, 384 here
        "down_blocks.0.attentions.0.transformer_blocks.0.attn2.to_v.weight": (320, 384),
        "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.0.proj.weight": (2560, 320),                                # GEGLU: 2 * 4 * C
        "down_blocks.0.attentions.0.transformer_blocks.0.ff.net.2.weight": (320, 1280),
        "down_blocks.0.downsamplers.0.conv.weight": (320, 320, 3, 3),
        "down_blocks.1.resnets.0.conv_shortcut.weight": (640, 320, 1, 1),
        "down_blocks.2.downsamplers.0.conv.weight": (1280, 1280, 3, 3),
        "down_blocks.3.resnets.1.conv2.weight": (1280, 1280, 3, 3),
        "mid_block.resnets.1.conv1.weight": (1280, 1280, 3, 3),
        "mid_block.attentions.0.transformer_blocks.0.attn1.to_q.weight": (1280, 1280),
        "up_blocks.0.resnets.0.conv1.weight": (1280, 2560, 3, 3), "up_blocks.0.resnets.0.conv_shortcut.weight": (1280, 2560, 1, 1),
        "up_blocks.0.upsamplers.0.conv.weight": (1280, 1280, 3, 3),
        "up_blocks.1.resnets.2.conv1.weight": (1280, 1920, 3, 3),                                                           # 1280 hidden + 640 skip
        "up_blocks.2.resnets.0.conv1.weight": (640, 1920, 3, 3), "up_blocks.2.resnets.2.conv1.weight": (640, 960, 3, 3),
        "up_blocks.3.resnets.0.conv1.weight": (320, 960, 3, 3), "up_blocks.3.resnets.2.conv1.weight": (320, 640, 3, 3),
        "up_blocks.3.attentions.2.proj_out.weight": (320, 320, 1, 1),
        "conv_norm_out.weight": (320,), "conv_out.weight": (4, 320, 3, 3), "conv_out.bias": (4,),
    }
    for k, shape in known.items():
        assert unet_manifest.get(k) == shape, (k, unet_manifest.get(k), shape)
    # what must NOT exist: the last down block and the first up block have no attention; attn1 / attn2 q, k, v carry no bias
    assert not any(k.startswith(("down_blocks.3.attentions", "up_blocks.0.attentions", "down_blocks.3.downsamplers", "up_blocks.3.upsamplers")) for k in unet_manifest)
    assert not any(re.search(r"attn[12]\.to_[qkv]\.bias$", k) for k in unet_manifest)


def test_vae_decoder_parameter_count_matches_public_sd_vae(vae_manifest):
    dec = {k: s for k, s in vae_manifest.items() if k.startswith("decoder.")}
    pq = {k: s for k, s in vae_manifest.items() if k.startswith("post_quant_conv.")}
    assert _numel(dec) == SD_VAE_DECODER_PARAMS and _numel(pq) == SD_VAE_POST_QUANT_PARAMS
    assert set(vae_manifest) == set(dec) | set(pq)
    known = {"decoder.conv_in.weight": (512, 4, 3, 3), "decoder.mid_block.attentions.0.group_norm.weight": (512,),
             "decoder.mid_block.attentions.0.to_q.weight": (512, 512), "decoder.up_blocks.0.resnets.2.conv2.weight": (512, 512, 3, 3),
             "decoder.up_blocks.1.upsamplers.0.conv.weight": (512, 512, 3, 3), "decoder.up_blocks.2.resnets.0.conv_shortcut.weight": (256, 512, 1, 1),
             "decoder.up_blocks.3.resnets.0.conv1.weight": (128, 256, 3, 3), "decoder.conv_norm_out.weight": (128,),
             "decoder.conv_out.weight": (3, 128, 3, 3), "post_quant_conv.weight": (4, 4, 1, 1)}
    for k, shape in known.items():
        assert vae_manifest.get(k) == shape, (k, vae_manifest.get(k), shape)
    assert not any(k.startswith("decoder.up_blocks.3.upsamplers") for k in vae_manifest)


def test_legacy_vae_checkpoint_names_map_onto_the_manifest(vae_manifest):
    """The published sd-vae-ft-mse file stores the mid-block attention under the OLD diffusers names and as 1 x 1 convolutions."""
    from mere_fusion_amd.musetalk.models.vae import remap_legacy_attention_keys
    legacy = {}
    for k, s in vae_manifest.items():
        for new, old in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            if f".attentions.0.{new}." in k:
                k = k.replace(f".attentions.0.{new}.", f".attentions.0.{old}.")
                s = s + (1, 1) if len(s) == 2 else s
        legacy[k] = torch.empty(s, device="meta")
    assert any(".query." in k for k in legacy)
    back = remap_legacy_attention_keys(legacy)
    assert set(back) == set(vae_manifest)
    for k, v in back.items():                                                  # same element count: the C loader checks numel, not rank
        assert v.numel() == torch.empty(vae_manifest[k], device="meta").numel(), k


# ---- the oracle's shared blocks against torch's own modules ----------------------------------------------------
def test_oracle_groupnorm_and_attention_match_torch_modules():
    from oracle import musetalk_ref as R
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 64, 9, 7, generator=g)
    gn = torch.nn.GroupNorm(32, 64, eps=1e-6)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(64, generator=g)); gn.bias.copy_(torch.randn(64, generator=g))
        sd = {"n.weight": gn.weight.detach(), "n.bias": gn.bias.detach()}
        assert torch.allclose(R._gn(sd, "n", x, 32, 1e-6), gn(x), atol=1e-6)
        q, k, v = (torch.randn(2, 8, 37, 40, generator=g) for _ in range(3))      # [B, heads, T, dh]
        want = torch.nn.functional.scaled_dot_product_attention(q, k, v)
        got = R.attention_core(q.transpose(1, 2).reshape(2, 37, 320), k.transpose(1, 2).reshape(2, 37, 320), v.transpose(1, 2).reshape(2, 37, 320), 8)
        assert torch.allclose(got, want.transpose(1, 2).reshape(2, 37, 320), atol=2e-6)
