"""ORACLE (test infrastructure, not product): CPU fp32 restatement of MuseTalk's UNet + VAE-decode step.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

PARITY UNPINNED.  The arithmetic of this stage lives in `diffusers` (requirements.txt:19, un-vendored,
unpinned, not installed here) and is configured by files the reference does not ship
(`./models/musetalk/musetalk.json`, `./models/sd-vae-ft-mse/config.json`, musetalk/utils/utils.py:67-73).
The reference only fixes the call seams, which ARE restated exactly:
  * PositionalEncoding                     musetalk/models/unet.py:12-27
  * unet.model(latents, timesteps=[0], encoder_hidden_states=audio).sample   musereal.py:59,105-107
  * VAE.decode_latents                     musetalk/models/vae.py:96-108  (1/scaling_factor, decode,
                                           /2+0.5 clamp, NHWC, *255 round uint8, RGB->BGR)
What sits between is the published algorithm of diffusers' `UNet2DConditionModel` (SD-1.x family:
CrossAttnDownBlock2D x3 + DownBlock2D, UNetMidBlock2DCrossAttn, UpBlock2D + CrossAttnUpBlock2D x3,
ResnetBlock2D, Transformer2DModel with one BasicTransformerBlock, GEGLU feed-forward, nearest-2x
upsampling, stride-2 downsampling convs, sinusoidal timestep embedding with flip_sin_to_cos) and
`AutoencoderKL`'s decoder (post_quant_conv, conv_in, mid resnet-attention-resnet, 4 up blocks of 3
resnets, GroupNorm/SiLU/conv_out), restated from its documentation with diffusers' state-dict key names so
real checkpoints map onto it [upstream-knowledge, SURVEY Appendix C].  The configuration is a parameter:
`MUSETALK_V1` is the assumed production config; tests also run a reduced one.

What pins it short of diffusers itself (tests/test_musetalk_manifest.py, tests/test_musetalk.py): the state-dict manifest it consumes
has the public SD-1.x UNet's 686 tensors and exactly 859,520,964 + 11,520 (in_channels 8) - 9,584,640 (cross-attention dim 384)
= 849,947,844 parameters, the decoder manifest sd-vae-ft-mse's 49,490,179, and a list of published (name, shape) entries; `_gn` /
`attention_core` agree with `torch.nn.GroupNorm` / `F.scaled_dot_product_attention`; the MAC count reproduces SURVEY Appendix C.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

MUSETALK_V1 = dict(
    unet=dict(in_channels=8, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
              cross_attention_dim=384, attention_heads=8, norm_num_groups=32,
              down_attn=(True, True, True, False), up_attn=(False, True, True, True)),
    vae=dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
             norm_num_groups=32, scaling_factor=0.18215),
)
# same topology, 1/8 of the width: CPU-second parity runs
MUSETALK_SMALL = dict(
    unet=dict(in_channels=8, out_channels=4, block_out_channels=(64, 128, 192, 192), layers_per_block=2,
              cross_attention_dim=384, attention_heads=8, norm_num_groups=32,
              down_attn=(True, True, True, False), up_attn=(False, True, True, True)),
    vae=dict(latent_channels=4, out_channels=3, block_out_channels=(32, 64, 128, 128), layers_per_block=2,
             norm_num_groups=32, scaling_factor=0.18215),
)


# ---- musetalk/models/unet.py:12-27 ------------------------------------------------------------------------
def positional_encoding(seq_len, d_model=384):
    pe = torch.zeros(seq_len, d_model)
    position = torch.arange(0, seq_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


def add_positional_encoding(x):
    return x + positional_encoding(x.shape[1], x.shape[2])[None]


# ---- shared blocks ---------------------------------------------------------------------------------------
def _gn(sd, p, x, groups, eps):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _lin(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _resnet(sd, p, x, temb_act, groups, eps):
    """ResnetBlock2D: norm1-silu-conv1 (+ time_emb_proj(silu(temb))) - norm2-silu-conv2, + (conv_)shortcut."""
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x, groups, eps)))
    if temb_act is not None and (p + ".time_emb_proj.weight") in sd:
        h = h + _lin(sd, p + ".time_emb_proj", temb_act)[:, :, None, None]
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h, groups, eps)))
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def attention_core(q, k, v, heads):
    """softmax(q k^T * dim_head**-0.5) v per head on [B, T, heads*dim_head] tensors (diffusers Attention /
    whisper model.py:82-93 qkv_attention, which splits the same scale as dim_head**-0.25 on q and k)."""
    B, T, C = q.shape
    dh = C // heads
    q = q.view(B, T, heads, dh).transpose(1, 2)
    k = k.view(B, -1, heads, dh).transpose(1, 2)
    v = v.view(B, -1, heads, dh).transpose(1, 2)
    w = torch.softmax((q @ k.transpose(-1, -2)) * dh ** -0.5, dim=-1)
    return (w @ v).transpose(1, 2).reshape(B, T, C)


def _attention(sd, p, x, ctx, heads):
    """diffusers Attention: to_q/to_k/to_v (no bias in the UNet), scale = dim_head**-0.5, to_out.0."""
    q, k, v = _lin(sd, p + ".to_q", x), _lin(sd, p + ".to_k", ctx), _lin(sd, p + ".to_v", ctx)
    return _lin(sd, p + ".to_out.0", attention_core(q, k, v, heads))


def _transformer(sd, p, x, ctx, heads, groups):
    """Transformer2DModel (conv projections) with one BasicTransformerBlock."""
    B, C, H, W_ = x.shape
    h = _conv(sd, p + ".proj_in", _gn(sd, p + ".norm", x, groups, 1e-6), padding=0)
    h = h.permute(0, 2, 3, 1).reshape(B, H * W_, C)
    t = p + ".transformer_blocks.0"
    n = F.layer_norm(h, (C,), sd[t + ".norm1.weight"], sd[t + ".norm1.bias"])
    h = h + _attention(sd, t + ".attn1", n, n, heads)
    n = F.layer_norm(h, (C,), sd[t + ".norm2.weight"], sd[t + ".norm2.bias"])
    h = h + _attention(sd, t + ".attn2", n, ctx, heads)
    n = F.layer_norm(h, (C,), sd[t + ".norm3.weight"], sd[t + ".norm3.bias"])
    a, gate = _lin(sd, t + ".ff.net.0.proj", n).chunk(2, dim=-1)       # GEGLU
    h = h + _lin(sd, t + ".ff.net.2", a * F.gelu(gate))
    h = h.reshape(B, H, W_, C).permute(0, 3, 1, 2)
    return _conv(sd, p + ".proj_out", h, padding=0) + x


def timestep_embedding(timesteps, dim):
    """get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


@torch.no_grad()
def unet_forward(sd, cfg, sample, timestep, encoder_hidden_states, taps=None):
    """UNet2DConditionModel.forward(sample, timestep, encoder_hidden_states).sample"""
    boc, G, heads = cfg["block_out_channels"], cfg["norm_num_groups"], cfg["attention_heads"]
    L = cfg["layers_per_block"]
    t = torch.as_tensor(timestep).reshape(-1).expand(sample.shape[0])
    temb = _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", timestep_embedding(t, boc[0]))))
    temb_act = F.silu(temb)
    x = _conv(sd, "conv_in", sample)
    skips = [x]
    for b in range(len(boc)):
        for i in range(L):
            x = _resnet(sd, f"down_blocks.{b}.resnets.{i}", x, temb_act, G, 1e-5)
            if cfg["down_attn"][b]:
                x = _transformer(sd, f"down_blocks.{b}.attentions.{i}", x, encoder_hidden_states, heads, G)
            skips.append(x)
        if b < len(boc) - 1:
            x = _conv(sd, f"down_blocks.{b}.downsamplers.0.conv", x, stride=2)
            skips.append(x)
    if taps is not None:
        taps["down"] = x
    x = _resnet(sd, "mid_block.resnets.0", x, temb_act, G, 1e-5)
    x = _transformer(sd, "mid_block.attentions.0", x, encoder_hidden_states, heads, G)
    x = _resnet(sd, "mid_block.resnets.1", x, temb_act, G, 1e-5)
    if taps is not None:
        taps["mid"] = x
    for b in range(len(boc)):
        for i in range(L + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = _resnet(sd, f"up_blocks.{b}.resnets.{i}", x, temb_act, G, 1e-5)
            if cfg["up_attn"][b]:
                x = _transformer(sd, f"up_blocks.{b}.attentions.{i}", x, encoder_hidden_states, heads, G)
        if b < len(boc) - 1:
            x = _conv(sd, f"up_blocks.{b}.upsamplers.0.conv", F.interpolate(x, scale_factor=2.0, mode="nearest"))
    x = F.silu(_gn(sd, "conv_norm_out", x, G, 1e-5))
    return _conv(sd, "conv_out", x)


def _vae_attention(sd, p, x, groups):
    """AutoencoderKL mid-block attention: one head over all channels, residual connection."""
    B, C, H, W_ = x.shape
    h = _gn(sd, p + ".group_norm", x, groups, 1e-6).view(B, C, H * W_).transpose(1, 2)
    q, k, v = _lin(sd, p + ".to_q", h), _lin(sd, p + ".to_k", h), _lin(sd, p + ".to_v", h)
    w = torch.softmax((q @ k.transpose(-1, -2)) * C ** -0.5, dim=-1)
    o = _lin(sd, p + ".to_out.0", w @ v)
    return o.transpose(1, 2).reshape(B, C, H, W_) + x


@torch.no_grad()
def vae_decode(sd, cfg, z, taps=None):
    """AutoencoderKL.decode(z).sample"""
    boc, G, L = cfg["block_out_channels"], cfg["norm_num_groups"], cfg["layers_per_block"]
    x = _conv(sd, "post_quant_conv", z, padding=0)
    x = _conv(sd, "decoder.conv_in", x)
    x = _resnet(sd, "decoder.mid_block.resnets.0", x, None, G, 1e-6)
    x = _vae_attention(sd, "decoder.mid_block.attentions.0", x, G)
    x = _resnet(sd, "decoder.mid_block.resnets.1", x, None, G, 1e-6)
    if taps is not None:
        taps["mid"] = x
    for b in range(len(boc)):
        for i in range(L + 1):
            x = _resnet(sd, f"decoder.up_blocks.{b}.resnets.{i}", x, None, G, 1e-6)
        if b < len(boc) - 1:
            x = _conv(sd, f"decoder.up_blocks.{b}.upsamplers.0.conv", F.interpolate(x, scale_factor=2.0, mode="nearest"))
    x = F.silu(_gn(sd, "decoder.conv_norm_out", x, G, 1e-6))
    return _conv(sd, "decoder.conv_out", x)


def decode_latents(vae_sd, cfg, latents):
    """musetalk/models/vae.py:96-108 -> uint8 (B, H, W, 3) BGR."""
    image = vae_decode(vae_sd, cfg, (1 / cfg["scaling_factor"]) * latents)
    image = (image / 2 + 0.5).clamp(0, 1)
    image = image.permute(0, 2, 3, 1).float().numpy()
    image = (image * 255).round().astype("uint8")
    return image[..., ::-1]


def musetalk_step(unet_sd, vae_sd, cfg, latent_batch, whisper_batch):
    """musereal.py:100-108: pe(audio) -> unet(latents, t=0, audio) -> vae.decode_latents."""
    audio = add_positional_encoding(torch.as_tensor(whisper_batch, dtype=torch.float32))
    pred = unet_forward(unet_sd, cfg["unet"], latent_batch, torch.tensor([0]), audio)
    return decode_latents(vae_sd, cfg["vae"], pred), pred


# ---- AutoencoderKL.encode: avatar preparation (musetalk/models/vae.py:52-94, 110-122) --------------------------------------------
def preprocess_img(img_bgr_u8, half_mask=False):
    """vae.py:52-82 for an in-memory crop: BGR uint8 [256, 256, 3] -> fp32 [1, 3, 256, 256] RGB, / 255., half mask, Normalize(.5, .5)."""
    x = np.asarray([np.asarray(img_bgr_u8)[:, :, ::-1]]) / 255.
    x = torch.squeeze(torch.FloatTensor(np.transpose(x, (3, 0, 1, 2))))
    if half_mask:
        m = torch.zeros((x.shape[1], x.shape[2]))
        m[:x.shape[1] // 2, :] = 1
        x = x * (m > 0.5)
    return ((x - 0.5) / 0.5).unsqueeze(0)


@torch.no_grad()
def vae_encode_moments(sd, cfg, image):
    """diffusers AutoencoderKL.encode up to the distribution parameters: Encoder (conv_in; per block 2 resnets + Downsample2D with
    F.pad(x, (0, 1, 0, 1)) and a stride-2 pad-0 conv; mid resnet - attention - resnet; GroupNorm, SiLU, conv_out) then quant_conv.
    Returns (mean | logvar) [B, 2 * latent, 32, 32]."""
    G, boc, L = cfg["norm_num_groups"], cfg["block_out_channels"], cfg["layers_per_block"]
    h = _conv(sd, "encoder.conv_in", image)
    for b in range(len(boc)):
        for i in range(L):
            h = _resnet(sd, f"encoder.down_blocks.{b}.resnets.{i}", h, None, G, 1e-6)
        if b < len(boc) - 1:
            h = _conv(sd, f"encoder.down_blocks.{b}.downsamplers.0.conv", F.pad(h, (0, 1, 0, 1)), stride=2, padding=0)
    h = _resnet(sd, "encoder.mid_block.resnets.0", h, None, G, 1e-6)
    h = _vae_attention(sd, "encoder.mid_block.attentions.0", h, G)
    h = _resnet(sd, "encoder.mid_block.resnets.1", h, None, G, 1e-6)
    h = _conv(sd, "encoder.conv_out", F.silu(_gn(sd, "encoder.conv_norm_out", h, G, 1e-6)))
    return _conv(sd, "quant_conv", h, padding=0)


def sample_latents(moments, scaling_factor, noise):
    """DiagonalGaussianDistribution.sample() with a given noise tensor, times the scaling factor (vae.py:92-93)."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return scaling_factor * (mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise)


def count_macs(cfg, hw=32, ctx_len=50):
    """Analytic conv/linear/attention MACs per frame of unet_forward + vae_decode (the FLOP numerator)."""
    u, v = cfg["unet"], cfg["vae"]
    boc, L = u["block_out_channels"], u["layers_per_block"]
    macs = {"unet": 0, "vae": 0}

    def conv(k, ci, co, s, key):
        macs[key] += ci * co * k * k * s * s

    def resnet(ci, co, s, key):
        conv(3, ci, co, s, key); conv(3, co, co, s, key)
        if ci != co:
            conv(1, ci, co, s, key)

    def xf(c, s, heads, key):
        T = s * s
        macs[key] += 2 * c * c * T          # proj_in, proj_out
        macs[key] += 4 * c * c * T          # attn1 q k v out
        macs[key] += 2 * T * T * c          # attn1 scores + pv
        macs[key] += 2 * c * c * T + 2 * u["cross_attention_dim"] * c * ctx_len   # attn2 q, out ; k, v
        macs[key] += 2 * T * ctx_len * c
        macs[key] += c * 8 * c * T + 4 * c * c * T   # GEGLU ff
    s = hw
    conv(3, u["in_channels"], boc[0], s, "unet")
    chans = [boc[0]]
    c = boc[0]
    for b, co in enumerate(boc):
        for _ in range(L):
            resnet(c, co, s, "unet"); c = co
            if u["down_attn"][b]:
                xf(c, s, u["attention_heads"], "unet")
            chans.append(c)
        if b < len(boc) - 1:
            s //= 2
            conv(3, c, c, s, "unet"); chans.append(c)
    resnet(c, c, s, "unet"); xf(c, s, u["attention_heads"], "unet"); resnet(c, c, s, "unet")
    for b, co in enumerate(reversed(boc)):
        for _ in range(L + 1):
            resnet(c + chans.pop(), co, s, "unet"); c = co
            if u["up_attn"][b]:
                xf(c, s, u["attention_heads"], "unet")
        if b < len(boc) - 1:
            s *= 2
            conv(3, c, c, s, "unet")
    conv(3, c, u["out_channels"], s, "unet")
    vb = v["block_out_channels"]
    s = hw
    macs["vae"] += v["latent_channels"] ** 2 * s * s
    c = vb[-1]
    conv(3, v["latent_channels"], c, s, "vae")
    resnet(c, c, s, "vae"); resnet(c, c, s, "vae")
    macs["vae"] += 4 * c * c * s * s + 2 * (s * s) ** 2 * c
    for b, co in enumerate(reversed(vb)):
        for _ in range(v["layers_per_block"] + 1):
            resnet(c, co, s, "vae"); c = co
        if b < len(vb) - 1:
            s *= 2
            conv(3, c, c, s, "vae")
    conv(3, c, v["out_channels"], s, "vae")
    return macs
