"""Seeded synthetic checkpoints for the Wav2Lip generator.

The reference ships no weights (`./models/wav2lip.pth` is absent, lipreal.py:76), so
benchmarks, parity tests and the golden fixtures all use the state dict produced here.
numpy's PCG64 stream is bit-reproducible across machines, so the same seed gives the same
352 tensors in this container (where the reference is imported to make goldens) and on the
GPU box (where it is not).

Key names and shapes follow the module tree of wav2lip/models/wav2lip.py:12-85 and
wav2lip/models/conv.py:5-44 (conv_block.0 = conv / convT, conv_block.1 = BatchNorm2d).
BatchNorm statistics are deliberately non-trivial: with default init BN is an identity and
the sigmoid output collapses to ~0.5, which pins nothing.
"""
import numpy as np
import torch

# (prefix, kind, cin, cout, kh, kw) ; kind: "conv" | "convT" | "plain" (nn.Conv2d without BN)
# ConvTranspose2d weights are [cin, cout, kh, kw]; Conv2d weights are [cout, cin, kh, kw].


def _face_encoder():
    cfg = [
        [(6, 16, 7)],
        [(16, 32, 3), (32, 32, 3), (32, 32, 3)],
        [(32, 64, 3), (64, 64, 3), (64, 64, 3), (64, 64, 3)],
        [(64, 128, 3), (128, 128, 3), (128, 128, 3)],
        [(128, 256, 3), (256, 256, 3), (256, 256, 3)],
        [(256, 512, 3), (512, 512, 3)],
        [(512, 512, 3), (512, 512, 1)],
    ]
    out = []
    for b, blk in enumerate(cfg):
        for i, (ci, co, k) in enumerate(blk):
            out.append((f"face_encoder_blocks.{b}.{i}", "conv", ci, co, k, k))
    return out


def _audio_encoder():
    cfg = [(1, 32, 3), (32, 32, 3), (32, 32, 3), (32, 64, 3), (64, 64, 3), (64, 64, 3),
           (64, 128, 3), (128, 128, 3), (128, 128, 3), (128, 256, 3), (256, 256, 3),
           (256, 512, 3), (512, 512, 1)]
    return [(f"audio_encoder.{i}", "conv", ci, co, k, k) for i, (ci, co, k) in enumerate(cfg)]


def _face_decoder():
    cfg = [
        [("conv", 512, 512, 1)],
        [("convT", 1024, 512, 3), ("conv", 512, 512, 3)],
        [("convT", 1024, 512, 3), ("conv", 512, 512, 3), ("conv", 512, 512, 3)],
        [("convT", 768, 384, 3), ("conv", 384, 384, 3), ("conv", 384, 384, 3)],
        [("convT", 512, 256, 3), ("conv", 256, 256, 3), ("conv", 256, 256, 3)],
        [("convT", 320, 128, 3), ("conv", 128, 128, 3), ("conv", 128, 128, 3)],
        [("convT", 160, 64, 3), ("conv", 64, 64, 3), ("conv", 64, 64, 3)],
    ]
    out = []
    for b, blk in enumerate(cfg):
        for i, (kind, ci, co, k) in enumerate(blk):
            out.append((f"face_decoder_blocks.{b}.{i}", kind, ci, co, k, k))
    return out


def wav2lip_layer_table():
    """All 51 conv layers in state-dict order of the reference constructor."""
    return (_face_encoder() + _audio_encoder() + _face_decoder()
            + [("output_block.0", "conv", 80, 32, 3, 3), ("output_block.1", "plain", 32, 3, 1, 1)])


# Residual layers (conv.py:17-18 adds the input back) in constructor order; a smaller gain on
# them keeps the activation scale O(1) through all 51 layers so no output saturates.
_RESIDUAL = {p for p, *_ in _face_encoder() if not p.endswith(".0") and not p.startswith("face_encoder_blocks.6")}
_RESIDUAL |= {f"audio_encoder.{i}" for i in (1, 2, 4, 5, 7, 8, 10)}
_RESIDUAL |= {p for p, k, *_ in _face_decoder() if k == "conv" and not p.endswith(".0")}
G_RES, G_PLAIN, G_CONVT, G_OUT = 0.4, 0.8, 0.8, 2.5


def make_wav2lip_state_dict(seed=0, dtype=torch.float32):
    """352 tensors: 101 weight, 101 bias, 50 x (running_mean, running_var, num_batches_tracked)."""
    rng = np.random.default_rng(seed)
    sd = {}

    def f(a):
        return torch.from_numpy(np.asarray(a, dtype=np.float32)).to(dtype)

    for prefix, kind, ci, co, kh, kw in wav2lip_layer_table():
        fan_in = ci * kh * kw
        if kind == "convT":
            # stride-2 transposed conv: an output pixel sees on average 9/4 of the 9 taps;
            # the stride-1 one on a 1x1 input sees exactly one tap.
            eff = ci if prefix == "face_decoder_blocks.1.0" else fan_in / 4.0
            w = rng.standard_normal((ci, co, kh, kw)) * (np.sqrt(2.0 / eff) * G_CONVT)
        else:
            g = G_RES if prefix in _RESIDUAL else (G_OUT if kind == "plain" else G_PLAIN)
            w = rng.standard_normal((co, ci, kh, kw)) * (np.sqrt(2.0 / fan_in) * g)
        b = rng.standard_normal(co) * 0.05
        if kind == "plain":
            sd[f"{prefix}.weight"] = f(w)
            sd[f"{prefix}.bias"] = f(b)
            continue
        sd[f"{prefix}.conv_block.0.weight"] = f(w)
        sd[f"{prefix}.conv_block.0.bias"] = f(b)
        sd[f"{prefix}.conv_block.1.weight"] = f(rng.uniform(0.7, 1.3, co))
        sd[f"{prefix}.conv_block.1.bias"] = f(rng.standard_normal(co) * 0.1)
        sd[f"{prefix}.conv_block.1.running_mean"] = f(rng.standard_normal(co) * 0.1)
        sd[f"{prefix}.conv_block.1.running_var"] = f(rng.uniform(0.6, 1.4, co))
        sd[f"{prefix}.conv_block.1.num_batches_tracked"] = torch.tensor(1000, dtype=torch.long)
    return sd


def make_lip_inputs(batch, seed=0):
    """Synthetic cfg-2 inputs: mel [B,1,80,16] ~U(-4,4); face [B,6,96,96] in [0,1]
    built the way lipreal.py:115-122 builds it (masked copy has rows >= 48 zeroed)."""
    rng = np.random.default_rng(1000 + seed)
    mel = rng.uniform(-4.0, 4.0, (batch, 1, 80, 16)).astype(np.float32)
    u8 = rng.integers(0, 256, (batch, 96, 96, 3), dtype=np.uint8)
    masked = u8.copy()
    masked[:, 48:] = 0
    face = (np.concatenate((masked, u8), axis=3) / 255.0).transpose(0, 3, 1, 2).astype(np.float32)
    return torch.from_numpy(mel), torch.from_numpy(np.ascontiguousarray(face)), u8


# ---- Whisper-tiny audio encoder (musetalk/whisper/whisper/model.py:131-171) --------------------------
# dims of the public "tiny" checkpoint; the reference reads them from the checkpoint (whisper/__init__.py:112)
WHISPER_TINY = dict(n_mels=80, n_audio_ctx=1500, n_audio_state=384, n_audio_head=6, n_audio_layer=4)


def make_whisper_encoder_state_dict(seed=0, dims=WHISPER_TINY):
    """Seeded encoder weights under the reference's key names (AudioEncoder's own state_dict, i.e. without
    the 'encoder.' prefix a full Whisper checkpoint carries).  positional_embedding is the model's own
    sinusoid buffer and is regenerated, not drawn."""
    rng = np.random.default_rng(7000 + seed)
    C, M, L = dims["n_audio_state"], dims["n_mels"], dims["n_audio_layer"]

    def f(a):
        return torch.from_numpy(np.asarray(a, dtype=np.float32))

    def lin(out_f, in_f, gain=1.0):
        return f(rng.standard_normal((out_f, in_f)) * (gain / np.sqrt(in_f)))

    sd = {
        "conv1.weight": f(rng.standard_normal((C, M, 3)) * (1.0 / np.sqrt(M * 3))),
        "conv1.bias": f(rng.standard_normal(C) * 0.05),
        "conv2.weight": f(rng.standard_normal((C, C, 3)) * (1.4 / np.sqrt(C * 3))),
        "conv2.bias": f(rng.standard_normal(C) * 0.05),
    }
    for i in range(L):
        p = f"blocks.{i}."
        sd[p + "attn.query.weight"] = lin(C, C, 1.5); sd[p + "attn.query.bias"] = f(rng.standard_normal(C) * 0.05)
        sd[p + "attn.key.weight"] = lin(C, C, 1.5)
        sd[p + "attn.value.weight"] = lin(C, C); sd[p + "attn.value.bias"] = f(rng.standard_normal(C) * 0.05)
        sd[p + "attn.out.weight"] = lin(C, C, 0.7); sd[p + "attn.out.bias"] = f(rng.standard_normal(C) * 0.05)
        sd[p + "attn_ln.weight"] = f(rng.uniform(0.8, 1.2, C)); sd[p + "attn_ln.bias"] = f(rng.standard_normal(C) * 0.05)
        sd[p + "mlp.0.weight"] = lin(4 * C, C); sd[p + "mlp.0.bias"] = f(rng.standard_normal(4 * C) * 0.05)
        sd[p + "mlp.2.weight"] = lin(C, 4 * C, 0.7); sd[p + "mlp.2.bias"] = f(rng.standard_normal(C) * 0.05)
        sd[p + "mlp_ln.weight"] = f(rng.uniform(0.8, 1.2, C)); sd[p + "mlp_ln.bias"] = f(rng.standard_normal(C) * 0.05)
    sd["ln_post.weight"] = f(rng.uniform(0.8, 1.2, C)); sd["ln_post.bias"] = f(rng.standard_normal(C) * 0.05)
    return sd


def make_speech_like_wav(n, seed=0):
    """Seeded 16 kHz test signal: band-limited noise bursts + a few tones, |x| <= 1 (SURVEY 8d cfg 3 feeds
    36x320 or 52x320 samples per step)."""
    rng = np.random.default_rng(9000 + seed)
    t = np.arange(n) / 16000.0
    x = 0.08 * rng.standard_normal(n)
    for f0 in (180.0, 440.0, 1250.0, 3100.0):
        x += 0.1 * np.sin(2 * np.pi * f0 * t + rng.uniform(0, 6.28)) * (0.5 + 0.5 * np.sin(2 * np.pi * rng.uniform(1, 4) * t))
    return np.clip(x, -1, 1).astype(np.float32)


# ---- MuseTalk UNet2DConditionModel / AutoencoderKL decoder (diffusers key names) -----
def _mt_gen(seed, shapes_only=False):
    """shapes_only: meta tensors (names + shapes, no storage) -- the key manifest of a checkpoint without drawing 850 M numbers."""
    rng = np.random.default_rng(seed)
    meta = lambda *shape: torch.empty(shape, dtype=torch.float32, device="meta")

    def f(a):
        return torch.from_numpy(np.asarray(a, dtype=np.float32))

    def conv(sd, p, ci, co, k, gain=1.0, bias=True):
        if shapes_only:
            sd[p + ".weight"] = meta(co, ci, k, k)
            if bias:
                sd[p + ".bias"] = meta(co)
            return
        # float32 draws: the full-size UNet has 860 M parameters
        sd[p + ".weight"] = torch.from_numpy(rng.standard_normal((co, ci, k, k), dtype=np.float32) * np.float32(gain / np.sqrt(ci * k * k)))
        if bias:
            sd[p + ".bias"] = f(rng.standard_normal(co) * 0.05)

    def lin(sd, p, ci, co, gain=1.0, bias=True):
        if shapes_only:
            sd[p + ".weight"] = meta(co, ci)
            if bias:
                sd[p + ".bias"] = meta(co)
            return
        sd[p + ".weight"] = torch.from_numpy(rng.standard_normal((co, ci), dtype=np.float32) * np.float32(gain / np.sqrt(ci)))
        if bias:
            sd[p + ".bias"] = f(rng.standard_normal(co) * 0.05)

    def norm(sd, p, c):
        if shapes_only:
            sd[p + ".weight"] = meta(c); sd[p + ".bias"] = meta(c)
            return
        sd[p + ".weight"] = f(rng.uniform(0.8, 1.2, c))
        sd[p + ".bias"] = f(rng.standard_normal(c) * 0.05)

    def resnet(sd, p, ci, co, temb=None):
        norm(sd, p + ".norm1", ci); conv(sd, p + ".conv1", ci, co, 3, 1.4)
        if temb:
            lin(sd, p + ".time_emb_proj", temb, co)
        norm(sd, p + ".norm2", co); conv(sd, p + ".conv2", co, co, 3, 0.7)
        if ci != co:
            conv(sd, p + ".conv_shortcut", ci, co, 1)

    return rng, f, conv, lin, norm, resnet


def make_musetalk_unet_state_dict(cfg, seed=0, shapes_only=False):
    u = cfg["unet"] if "unet" in cfg else cfg
    rng, f, conv, lin, norm, resnet = _mt_gen(11000 + seed, shapes_only)
    boc, L, X = u["block_out_channels"], u["layers_per_block"], u["cross_attention_dim"]
    temb = boc[0] * 4
    sd = {}

    def xf(p, c):
        norm(sd, p + ".norm", c); conv(sd, p + ".proj_in", c, c, 1)
        t = p + ".transformer_blocks.0"
        norm(sd, t + ".norm1", c)
        for n_ in ("to_q", "to_k", "to_v"):
            lin(sd, t + ".attn1." + n_, c, c, 1.3, bias=False)
        lin(sd, t + ".attn1.to_out.0", c, c, 0.7)
        norm(sd, t + ".norm2", c)
        lin(sd, t + ".attn2.to_q", c, c, 1.3, bias=False)
        lin(sd, t + ".attn2.to_k", X, c, 1.3, bias=False); lin(sd, t + ".attn2.to_v", X, c, 1.0, bias=False)
        lin(sd, t + ".attn2.to_out.0", c, c, 0.7)
        norm(sd, t + ".norm3", c)
        lin(sd, t + ".ff.net.0.proj", c, 8 * c); lin(sd, t + ".ff.net.2", 4 * c, c, 0.7)
        conv(sd, p + ".proj_out", c, c, 1, 0.7)

    lin(sd, "time_embedding.linear_1", boc[0], temb); lin(sd, "time_embedding.linear_2", temb, temb)
    conv(sd, "conv_in", u["in_channels"], boc[0], 3)
    chans, c = [boc[0]], boc[0]
    for b, co in enumerate(boc):
        for i in range(L):
            resnet(sd, f"down_blocks.{b}.resnets.{i}", c, co, temb); c = co
            if u["down_attn"][b]:
                xf(f"down_blocks.{b}.attentions.{i}", c)
            chans.append(c)
        if b < len(boc) - 1:
            conv(sd, f"down_blocks.{b}.downsamplers.0.conv", c, c, 3); chans.append(c)
    resnet(sd, "mid_block.resnets.0", c, c, temb); xf("mid_block.attentions.0", c); resnet(sd, "mid_block.resnets.1", c, c, temb)
    for b, co in enumerate(reversed(boc)):
        for i in range(L + 1):
            resnet(sd, f"up_blocks.{b}.resnets.{i}", c + chans.pop(), co, temb); c = co
            if u["up_attn"][b]:
                xf(f"up_blocks.{b}.attentions.{i}", c)
        if b < len(boc) - 1:
            conv(sd, f"up_blocks.{b}.upsamplers.0.conv", c, c, 3)
    norm(sd, "conv_norm_out", c); conv(sd, "conv_out", c, u["out_channels"], 3, 0.5)
    return sd


def make_musetalk_vae_state_dict(cfg, seed=0, shapes_only=False):
    v = cfg["vae"] if "vae" in cfg else cfg
    rng, f, conv, lin, norm, resnet = _mt_gen(12000 + seed, shapes_only)
    boc, L, Z = v["block_out_channels"], v["layers_per_block"], v["latent_channels"]
    sd = {}
    conv(sd, "post_quant_conv", Z, Z, 1, 2.0)
    c = boc[-1]
    conv(sd, "decoder.conv_in", Z, c, 3)
    resnet(sd, "decoder.mid_block.resnets.0", c, c)
    a = "decoder.mid_block.attentions.0"
    norm(sd, a + ".group_norm", c)
    for n_ in ("to_q", "to_k", "to_v"):
        lin(sd, a + "." + n_, c, c, 1.3)
    lin(sd, a + ".to_out.0", c, c, 0.7)
    resnet(sd, "decoder.mid_block.resnets.1", c, c)
    for b, co in enumerate(reversed(boc)):
        for i in range(L + 1):
            resnet(sd, f"decoder.up_blocks.{b}.resnets.{i}", c, co); c = co
        if b < len(boc) - 1:
            conv(sd, f"decoder.up_blocks.{b}.upsamplers.0.conv", c, c, 3)
    norm(sd, "decoder.conv_norm_out", c); conv(sd, "decoder.conv_out", c, v["out_channels"], 3, 1.2)
    return sd


def make_musetalk_vae_encoder_state_dict(cfg, seed=0, shapes_only=False):
    """`encoder.*` + `quant_conv.*` of diffusers AutoencoderKL (avatar preparation, musetalk/models/vae.py:84-94)."""
    v = cfg["vae"] if "vae" in cfg else cfg
    rng, f, conv, lin, norm, resnet = _mt_gen(12500 + seed, shapes_only)
    boc, L, Z = v["block_out_channels"], v["layers_per_block"], v["latent_channels"]
    sd = {}
    conv(sd, "encoder.conv_in", v["out_channels"], boc[0], 3)
    c = boc[0]
    for b, co in enumerate(boc):
        for i in range(L):
            resnet(sd, f"encoder.down_blocks.{b}.resnets.{i}", c, co); c = co
        if b < len(boc) - 1:
            conv(sd, f"encoder.down_blocks.{b}.downsamplers.0.conv", c, c, 3)
    resnet(sd, "encoder.mid_block.resnets.0", c, c)
    a = "encoder.mid_block.attentions.0"
    norm(sd, a + ".group_norm", c)
    for n_ in ("to_q", "to_k", "to_v"):
        lin(sd, a + "." + n_, c, c, 1.3)
    lin(sd, a + ".to_out.0", c, c, 0.7)
    resnet(sd, "encoder.mid_block.resnets.1", c, c)
    norm(sd, "encoder.conv_norm_out", c); conv(sd, "encoder.conv_out", c, 2 * Z, 3, 0.6)
    conv(sd, "quant_conv", 2 * Z, 2 * Z, 1, 1.0)
    return sd


def make_musetalk_inputs(batch, seed=0, hw=32):
    """cfg-3 inputs: latents [B,8,hw,hw] ~ N(0,1)*0.18215-scaled pairs (vae.py:117-121),
    whisper chunks [B,50,384] ~ N(0,1) before the positional encoding."""
    rng = np.random.default_rng(13000 + seed)
    lat = (rng.standard_normal((batch, 8, hw, hw)) * 0.9).astype(np.float32)
    aud = rng.standard_normal((batch, 50, 384)).astype(np.float32)
    return torch.from_numpy(lat), torch.from_numpy(aud)


# ---- ER-NeRF radiance field (ernerf/nerf_triplane/network.py:93-160) -------------------------------------------------
def make_ernerf_field_state_dict(n_embeddings, seed=0, individual_dim=4, exp_eye=True):
    """Seeded stand-in for a trained `NeRFNetwork` checkpoint (missing from the reference checkout, .MISSING_LARGE_BLOBS):
    the tensors of the inference field with trained-like magnitudes -- grid features O(1) (a fresh GridEncoder starts at
    1e-4, grid.py:127-129, which would mute every downstream layer), He-scaled bias-free Linears (network.py:79)."""
    import torch
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    sd = {}
    for plane in ("xy", "yz", "xz"):
        sd[f"encoder_{plane}.embeddings"] = torch.from_numpy(rng.uniform(-1.0, 1.0, (n_embeddings, 1)).astype(np.float32))
    sig_in = 36 + 32 + (1 if exp_eye else 0)

    def lin(name, dims):
        for i, (o, c) in enumerate(dims):
            sd[f"{name}.net.{i}.weight"] = torch.from_numpy((rng.standard_normal((o, c)) * np.sqrt(2.0 / c)).astype(np.float32))
    lin("sigma_net", [(64, sig_in), (64, 64), (65, 64)])
    lin("color_net", [(64, 16 + 64 + individual_dim), (3, 64)])
    lin("aud_ch_att_net", [(64, 36), (32, 64)])
    lin("eye_att_net", [(16, 36), (1, 16)])
    return sd


def make_ernerf_sphere_bitfield(H=128, radius=0.45):
    """Synthetic `density_bitfield` (renderer.py:113): occupancy of a ball at the origin, one bit per voxel in the Morton order
    `march_rays` reads (raymarching.cu:56-71, 889-890), cascade 0 only."""
    i = np.arange(H, dtype=np.uint32)
    x, y, z = np.meshgrid(i, i, i, indexing="ij")
    c = (np.stack([x, y, z], -1).astype(np.float32) + 0.5) / H * 2 - 1
    occ = (np.linalg.norm(c, axis=-1) < radius).reshape(-1)

    def expand(v):
        v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
        v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
        v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
        v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
        return v
    m = (expand(x.reshape(-1)) | (expand(y.reshape(-1)) << 1) | (expand(z.reshape(-1)) << 2)).astype(np.int64)
    bits = np.zeros(H ** 3, np.uint8)
    bits[m] = occ
    return np.packbits(bits, bitorder="little")


def make_ernerf_camera_rays(W):
    """W x W pinhole rays looking down +z at the unit box from z = -2.2 (what `get_rays`, utils.py, produces for a frontal pose)."""
    u = (np.arange(W, dtype=np.float32) + 0.5) / W * 2 - 1
    px, py = np.meshgrid(u, u)
    d = np.stack([px * 0.35, py * 0.35, np.ones_like(px)], -1).reshape(-1, 3)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    o = np.tile(np.array([[0.02, -0.01, -2.2]], np.float32), (W * W, 1))
    return o, d


def make_ernerf_audio_state_dict(template, seed=0):
    """Seeded tensors for `audio_net.*` / `audio_att_net.*` (network.py:9-66) with the shapes of `template` (a state dict of the
    reference module or of a fixture): He-scaled weights, small biases, from numpy PCG64 so they do not depend on torch's RNG."""
    import torch
    rng = np.random.Generator(np.random.PCG64(2000 + seed))
    sd = {}
    for k in sorted(template):
        if not k.startswith(("audio_net.", "audio_att_net.")):
            continue
        shape = tuple(template[k].shape)
        if k.endswith("weight"):
            fan_in = int(np.prod(shape[1:]))
            sd[k] = torch.from_numpy((rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32))
        else:
            sd[k] = torch.from_numpy((rng.standard_normal(shape) * 0.05).astype(np.float32))
    return sd


def make_ernerf_torso_state_dict(n_embeddings, seed=0, individual_dim=8, grid_size=128):
    """Seeded stand-in for the torso tensors of a trained NeRFNetwork(torso=True) (network.py:146-159): He-scaled deform / colour
    MLPs (the deform net's last layer scaled down so |dx| stays a few percent of the image), O(1) tiled-grid features, a smooth
    blob as `density_grid_torso`, the default anchor points, small individual codes."""
    import torch
    rng = np.random.Generator(np.random.PCG64(3000 + seed))
    sd = {"torso_encoder.embeddings": torch.from_numpy(rng.uniform(-1, 1, (n_embeddings, 2)).astype(np.float32))}
    din = 34 + 42 + individual_dim

    def lin(name, dims, last_scale=1.0):
        for i, (o, c) in enumerate(dims):
            w = rng.standard_normal((o, c)) * np.sqrt(2.0 / c) * (last_scale if i == len(dims) - 1 else 1.0)
            sd[f"{name}.net.{i}.weight"] = torch.from_numpy(w.astype(np.float32))
    lin("torso_deform_net", [(32, din), (32, 32), (2, 32)], 0.03)
    lin("torso_net", [(32, 32 + din), (32, 32), (4, 32)])
    u = (np.arange(grid_size, dtype=np.float32) + 0.5) / grid_size * 2 - 1
    yy, xx = np.meshgrid(u, u, indexing="ij")
    blob = np.clip(1.2 - np.sqrt((xx / 0.7) ** 2 + ((yy - 0.45) / 0.5) ** 2) * 1.5, 0, None)
    sd["density_grid_torso"] = torch.from_numpy(blob.astype(np.float32).reshape(-1))
    sd["anchor_points"] = torch.tensor([[0.01, 0.01, 0.1, 1], [-0.1, -0.1, 0.1, 1], [0.1, -0.1, 0.1, 1]], dtype=torch.float32)   # network.py:149
    if individual_dim:
        sd["individual_codes_torso"] = torch.from_numpy((rng.standard_normal((4, individual_dim)) * 0.1).astype(np.float32))
    return sd


# ---- wav2vec2 / HuBERT CTC network of NerfASR (nerfasr.py:41-45; transformers' Wav2Vec2ForCTC key names) ---------------------------------
# cpierse/wav2vec2-large-xlsr-53-esperanto (app.py:660) is the XLSR-53 "large" architecture with a 44-symbol head (nerfasr.py:19-20: audio_dim 44)
WAV2VEC2_XLSR_LARGE = dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096, vocab_size=44,
                           conv_dim=(512,) * 7, conv_kernel=(10, 3, 3, 3, 3, 2, 2), conv_stride=(5, 2, 2, 2, 2, 2, 2), conv_bias=True,
                           feat_extract_norm="layer", do_stable_layer_norm=True, num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16,
                           layer_norm_eps=1e-5)
# reduced twin for quick tests: same structure, head dim 64 (the fused attention kernel's), 2 layers
WAV2VEC2_SMALL = dict(WAV2VEC2_XLSR_LARGE, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, vocab_size=44,
                      conv_dim=(64,) * 7, num_conv_pos_embedding_groups=2)


def make_wav2vec2_state_dict(cfg, seed=0, shapes_only=False):
    """Seeded Wav2Vec2ForCTC weights under transformers' own state-dict names (the positional conv in torch's parametrised weight-norm form).
    shapes_only: {name: shape} without drawing anything (manifest tests)."""
    rng = np.random.default_rng(12000 + seed)
    C, F, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    sd = {}

    def put(name, shape, scale=None, kind="normal"):
        if shapes_only:
            sd[name] = tuple(shape)
            return
        if kind == "gamma":
            a = rng.uniform(0.8, 1.2, shape)
        else:
            a = rng.standard_normal(shape) * scale
        sd[name] = torch.from_numpy(np.asarray(a, dtype=np.float32))

    cin = 1
    for i, (co, k) in enumerate(zip(cfg["conv_dim"], cfg["conv_kernel"])):
        p = f"wav2vec2.feature_extractor.conv_layers.{i}."
        put(p + "conv.weight", (co, cin, k), 1.4 / np.sqrt(cin * k))
        if cfg["conv_bias"]:
            put(p + "conv.bias", (co,), 0.05)
        put(p + "layer_norm.weight", (co,), kind="gamma"); put(p + "layer_norm.bias", (co,), 0.05)
        cin = co
    put("wav2vec2.masked_spec_embed", (C,), 1.0)                                   # training-only parameter of the real checkpoints
    put("wav2vec2.feature_projection.layer_norm.weight", (cin,), kind="gamma"); put("wav2vec2.feature_projection.layer_norm.bias", (cin,), 0.05)
    put("wav2vec2.feature_projection.projection.weight", (C, cin), 1.0 / np.sqrt(cin)); put("wav2vec2.feature_projection.projection.bias", (C,), 0.05)
    K, G = cfg["num_conv_pos_embeddings"], cfg["num_conv_pos_embedding_groups"]
    pc = "wav2vec2.encoder.pos_conv_embed.conv."
    put(pc + "bias", (C,), 0.05)
    put(pc + "parametrizations.weight.original0", (1, 1, K), 0.0)                  # g, set below
    put(pc + "parametrizations.weight.original1", (C, C // G, K), 1.0)             # v
    if not shapes_only:
        # g[k] = gain * ||v[:, :, k]||: an effective weight of std gain / sqrt(fan_in)
        v = sd[pc + "parametrizations.weight.original1"].numpy()
        nrm = np.sqrt((v.astype(np.float64) ** 2).sum((0, 1)))
        sd[pc + "parametrizations.weight.original0"] = torch.from_numpy((nrm * (0.7 / np.sqrt(C // G * K)) * rng.uniform(0.8, 1.2, K)).astype(np.float32).reshape(1, 1, K))
    put("wav2vec2.encoder.layer_norm.weight", (C,), kind="gamma"); put("wav2vec2.encoder.layer_norm.bias", (C,), 0.05)
    for l in range(cfg["num_hidden_layers"]):
        p = f"wav2vec2.encoder.layers.{l}."
        for n, g in (("q_proj", 1.5), ("k_proj", 1.5), ("v_proj", 1.0), ("out_proj", 0.7)):
            put(p + f"attention.{n}.weight", (C, C), g / np.sqrt(C)); put(p + f"attention.{n}.bias", (C,), 0.05)
        put(p + "layer_norm.weight", (C,), kind="gamma"); put(p + "layer_norm.bias", (C,), 0.05)
        put(p + "feed_forward.intermediate_dense.weight", (F, C), 1.0 / np.sqrt(C)); put(p + "feed_forward.intermediate_dense.bias", (F,), 0.05)
        put(p + "feed_forward.output_dense.weight", (C, F), 0.7 / np.sqrt(F)); put(p + "feed_forward.output_dense.bias", (C,), 0.05)
        put(p + "final_layer_norm.weight", (C,), kind="gamma"); put(p + "final_layer_norm.bias", (C,), 0.05)
    put("lm_head.weight", (V, C), 1.0 / np.sqrt(C)); put("lm_head.bias", (V,), 0.05)
    return sd


# ---- avatar preparation: S3FD (net_s3fd.py:22-70) and BiSeNet (face_parsing/model.py + resnet.py) under the reference's module names -------
_S3FD_CONVS = [("conv1_1", 3, 64, 3), ("conv1_2", 64, 64, 3), ("conv2_1", 64, 128, 3), ("conv2_2", 128, 128, 3), ("conv3_1", 128, 256, 3),
               ("conv3_2", 256, 256, 3), ("conv3_3", 256, 256, 3), ("conv4_1", 256, 512, 3), ("conv4_2", 512, 512, 3), ("conv4_3", 512, 512, 3),
               ("conv5_1", 512, 512, 3), ("conv5_2", 512, 512, 3), ("conv5_3", 512, 512, 3), ("fc6", 512, 1024, 3), ("fc7", 1024, 1024, 1),
               ("conv6_1", 1024, 256, 1), ("conv6_2", 256, 512, 3), ("conv7_1", 512, 128, 1), ("conv7_2", 128, 256, 3),
               ("conv3_3_norm_mbox_conf", 256, 4, 3), ("conv3_3_norm_mbox_loc", 256, 4, 3), ("conv4_3_norm_mbox_conf", 512, 2, 3),
               ("conv4_3_norm_mbox_loc", 512, 4, 3), ("conv5_3_norm_mbox_conf", 512, 2, 3), ("conv5_3_norm_mbox_loc", 512, 4, 3),
               ("fc7_mbox_conf", 1024, 2, 3), ("fc7_mbox_loc", 1024, 4, 3), ("conv6_2_mbox_conf", 512, 2, 3), ("conv6_2_mbox_loc", 512, 4, 3),
               ("conv7_2_mbox_conf", 256, 2, 3), ("conv7_2_mbox_loc", 256, 4, 3)]


def make_s3fd_state_dict(seed=0, shapes_only=False):
    rng = np.random.default_rng(14000 + seed)
    sd = {}

    def put(name, shape, scale):
        sd[name] = tuple(shape) if shapes_only else torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))

    for name, ci, co, k in _S3FD_CONVS:
        head = "mbox" in name
        put(name + ".weight", (co, ci, k, k), (0.3 if head else 1.0) * np.sqrt(2.0 / (ci * k * k)))
        put(name + ".bias", (co,), 0.3 if head else 0.05)
    for name, ch, scale in (("conv3_3_norm", 256, 10.0), ("conv4_3_norm", 512, 8.0), ("conv5_3_norm", 512, 5.0)):
        sd[name + ".weight"] = (ch,) if shapes_only else torch.from_numpy((scale * rng.uniform(0.8, 1.2, ch)).astype(np.float32))
    return sd


def make_bisenet_state_dict(seed=0, n_classes=19, shapes_only=False):
    rng = np.random.default_rng(15000 + seed)
    sd = {}

    def conv(name, co, ci, k, gain=1.0):
        sd[name] = (co, ci, k, k) if shapes_only else torch.from_numpy((rng.standard_normal((co, ci, k, k)) * gain * np.sqrt(2.0 / (ci * k * k))).astype(np.float32))

    def bn(p, ch):
        for key, gen in (("weight", lambda: rng.uniform(0.8, 1.2, ch)), ("bias", lambda: rng.standard_normal(ch) * 0.1),
                         ("running_mean", lambda: rng.standard_normal(ch) * 0.1), ("running_var", lambda: rng.uniform(0.5, 1.5, ch))):
            sd[f"{p}.{key}"] = (ch,) if shapes_only else torch.from_numpy(gen().astype(np.float32))

    def cbr(p, ci, co, k):
        conv(p + ".conv.weight", co, ci, k); bn(p + ".bn", co)

    conv("cp.resnet.conv1.weight", 64, 3, 7); bn("cp.resnet.bn1", 64)
    cin = 64
    for li, (co, stride) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2)), 1):
        for bi in range(2):
            p = f"cp.resnet.layer{li}.{bi}"
            conv(p + ".conv1.weight", co, cin, 3); bn(p + ".bn1", co)
            conv(p + ".conv2.weight", co, co, 3, 0.7); bn(p + ".bn2", co)
            if bi == 0 and (cin != co or stride != 1):
                conv(p + ".downsample.0.weight", co, cin, 1); bn(p + ".downsample.1", co)
            cin = co
    for p, ci in (("cp.arm16", 256), ("cp.arm32", 512)):
        cbr(p + ".conv", ci, 128, 3)
        conv(p + ".conv_atten.weight", 128, 128, 1); bn(p + ".bn_atten", 128)
    cbr("cp.conv_head32", 128, 128, 3); cbr("cp.conv_head16", 128, 128, 3); cbr("cp.conv_avg", 512, 128, 1)
    cbr("ffm.convblk", 256, 256, 1)
    conv("ffm.conv1.weight", 64, 256, 1); conv("ffm.conv2.weight", 256, 64, 1)
    for p, ci, mid in (("conv_out", 256, 256), ("conv_out16", 128, 64), ("conv_out32", 128, 64)):
        cbr(p + ".conv", ci, mid, 3)
        conv(p + ".conv_out.weight", n_classes, mid, 1, 2.0)
    return sd
