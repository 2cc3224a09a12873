"""ORACLE (test infrastructure, not product): torch fp32 restatement of the ER-NeRF radiance field.

Follows ernerf/nerf_triplane/network.py statement by statement: `encode_x` (:211-219, `split_xyz` :204-208), `density`
(:280-308), `forward` (:249-277), `MLP.forward` (:82-90, bias-free Linears with ReLU between them).  The grid / SH encoders
are the plain-C restatements of oracle/ernerf_ref.c, called exactly as `GridEncoder.forward` (grid.py:139-154) and
`SHEncoder.forward` (sphere_harmonics.py:75-86) call their extension.

PINNED above the extension boundary: tests/golden/make_ernerf_golden.py runs the reference's own `NeRFNetwork.forward` on the CPU of
the build container (its CUDA extensions replaced by oracle/ernerf_ref.c) and tests/test_ernerf.py checks this restatement against
that fixture (sigma rtol 2e-5, colour 2e-6).  The extension kernels themselves stay unpinned (see ernerf_ref.c).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import ctypes as C
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _clib():
    return C.CDLL(os.path.join(HERE, "libernerfref.so"))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def grid_encode(x01, emb, offsets, S, H, gridtype=0, align=0):
    """`_grid_encode.forward` (grid.py:19-58): [B, D] in [0, 1] -> [B, L*C]."""
    x01 = np.ascontiguousarray(x01, np.float32); emb = np.ascontiguousarray(emb, np.float32); offsets = np.ascontiguousarray(offsets, np.int32)
    B, D = x01.shape
    L, Cc = offsets.shape[0] - 1, emb.shape[1]
    out = np.zeros((L, B, Cc), np.float32)
    _clib().ref_grid_encode_forward(_p(x01), _p(emb), _p(offsets), _p(out), C.c_uint32(B), C.c_uint32(D), C.c_uint32(Cc), C.c_uint32(L),
                                    C.c_float(S), C.c_uint32(H), C.c_uint32(gridtype), C.c_int(align))
    return out.transpose(1, 0, 2).reshape(B, L * Cc)             # grid.py:52


def sh_encode(d, degree=4):
    d = np.ascontiguousarray(d, np.float32)
    out = np.zeros((d.shape[0], degree * degree), np.float32)
    _clib().ref_sh_encode_forward(_p(d), _p(out), C.c_uint32(d.shape[0]), C.c_uint32(degree))
    return out


def mlp(sd, name, x):
    n = sum(1 for k in sd if k.startswith(name + ".net.") and k.endswith(".weight"))
    for l in range(n):
        x = x @ sd[f"{name}.net.{l}.weight"].t()
        if l != n - 1:
            x = torch.relu(x)
    return x


def field_forward(sd, x, d, enc_a, c, e, offsets, S, H=64, bound=1.0):
    """NeRFNetwork.forward (network.py:249-277) in test mode.  x, d: [M, 3] tensors; enc_a [1, 32]; c [1, ind] or None; e [1, 1] or None."""
    xn = x.numpy().astype(np.float32)
    planes = (xn[:, :2], xn[:, 1:], np.concatenate([xn[:, :1], xn[:, -1:]], -1))      # split_xyz
    feats = []
    for name, pl in zip(("xy", "yz", "xz"), planes):
        x01 = ((pl + np.float32(bound)) / np.float32(2 * bound)).astype(np.float32)   # grid.py:144
        feats.append(grid_encode(x01, sd[f"encoder_{name}.embeddings"].numpy(), offsets, S, H))
    enc_x = torch.from_numpy(np.concatenate(feats, -1))
    aud_ch_att = mlp(sd, "aud_ch_att_net", enc_x)
    enc_w = enc_a.repeat(enc_x.shape[0], 1) * aud_ch_att
    if e is not None:
        eye_att = torch.sigmoid(mlp(sd, "eye_att_net", enc_x))
        h = torch.cat([enc_x, enc_w, e * eye_att], -1)
    else:
        eye_att = torch.zeros(enc_x.shape[0], 1)
        h = torch.cat([enc_x, enc_w], -1)
    h = mlp(sd, "sigma_net", h)
    sigma = torch.exp(h[..., 0])
    geo_feat = h[..., 1:]
    enc_d = torch.from_numpy(sh_encode(d.numpy()))
    hc = torch.cat([enc_d, geo_feat] + ([c.repeat(x.shape[0], 1)] if c is not None else []), -1)
    color = torch.sigmoid(mlp(sd, "color_net", hc)) * (1 + 2 * 0.001) - 0.001
    unc = torch.log(1 + torch.exp(torch.zeros(x.shape[0], 1)))
    return sigma, color, aud_ch_att.norm(dim=-1, keepdim=True), eye_att, unc


def encode_audio(sd, a, att=2):
    """NeRFNetwork.encode_audio (network.py:222-237) with emb off: AudioNet (:40-66) on every window, AudioAttNet (:9-36) over the 8."""
    F = torch.nn.functional
    x = a[:, :, 8 - 8:8 + 8]                                                    # win_size 16
    for i in (0, 2, 4, 6):
        x = F.leaky_relu(F.conv1d(x, sd[f"audio_net.encoder_conv.{i}.weight"], sd[f"audio_net.encoder_conv.{i}.bias"], stride=2, padding=1), 0.02)
    x = x.squeeze(-1)
    x = F.leaky_relu(F.linear(x, sd["audio_net.encoder_fc1.0.weight"], sd["audio_net.encoder_fc1.0.bias"]), 0.02)
    x = F.linear(x, sd["audio_net.encoder_fc1.2.weight"], sd["audio_net.encoder_fc1.2.bias"])          # [n, 32]
    if att <= 0:
        return x
    x = x.unsqueeze(0)                                                         # [1, 8, 32]
    y = x.permute(0, 2, 1)
    for i in (0, 2, 4, 6, 8):
        y = F.leaky_relu(F.conv1d(y, sd[f"audio_att_net.attentionConvNet.{i}.weight"], sd[f"audio_att_net.attentionConvNet.{i}.bias"], padding=1), 0.02)
    y = torch.softmax(F.linear(y.view(1, 8), sd["audio_att_net.attentionNet.0.weight"], sd["audio_att_net.attentionNet.0.bias"]), dim=1).view(1, 8, 1)
    return torch.sum(y * x, dim=1)


def freq_encode(x, degree):
    """`_freq_encoder.forward` (freq.py:19-33 -> kernel_freq) via the C restatement: [B, D] -> [B, D + 2 D degree]."""
    x = np.ascontiguousarray(x, np.float32)
    B, D = x.shape
    Cc = D + 2 * D * degree
    out = np.zeros((B, Cc), np.float32)
    _clib().ref_freq_encode_forward(_p(x), C.c_uint32(B), C.c_uint32(D), C.c_uint32(degree), C.c_uint32(Cc), _p(out))
    return out


def run_torso(sd, bg_coords, poses, bg_color, offsets, S, torso_shrink=0.8, thresh=0.0, grid_size=128, H=16):
    """`NeRFRenderer.run_torso` (renderer.py:294-352) + `forward_torso` (network.py:166-201), test mode (individual code 0)."""
    xy = torch.as_tensor(bg_coords, dtype=torch.float32).reshape(-1, 2)
    N = xy.shape[0]
    bg = torch.as_tensor(bg_color, dtype=torch.float32)
    bg = bg.expand(N, 3) if bg.dim() else bg
    occ = torch.nn.functional.grid_sample(sd["density_grid_torso"].view(1, 1, grid_size, grid_size), xy.view(1, -1, 1, 2), align_corners=True).view(-1)
    mask = occ > thresh
    alpha, color = torch.zeros(N, 1), torch.zeros(N, 3)
    deform = torch.zeros(N, 2)
    if mask.any():
        x = xy[mask] * torso_shrink
        c = sd["individual_codes_torso"][0:1] if "individual_codes_torso" in sd else None
        wa = sd["anchor_points"][None, ...] @ torch.as_tensor(poses, dtype=torch.float32).reshape(1, 4, 4).permute(0, 2, 1).inverse()
        wa = (wa[:, :, :2] / wa[:, :, 3, None] / wa[:, :, 2, None]).view(1, -1)
        enc_anchor = torch.from_numpy(freq_encode(wa.numpy(), 3))
        enc_x = torch.from_numpy(freq_encode(x.numpy(), 8))
        parts = [enc_x, enc_anchor.repeat(x.shape[0], 1)] + ([c.repeat(x.shape[0], 1)] if c is not None else [])
        h = torch.cat(parts, -1)
        dx = mlp(sd, "torso_deform_net", h)
        x2 = (x + dx).clamp(-1, 1)
        g = torch.from_numpy(grid_encode(((x2.numpy() + np.float32(1)) / np.float32(2)).astype(np.float32), sd["torso_encoder.embeddings"].numpy(), offsets, S, H,
                                         gridtype=1))
        o = mlp(sd, "torso_net", torch.cat([g, h], -1))
        alpha[mask] = torch.sigmoid(o[..., :1]) * (1 + 2 * 0.001) - 0.001
        color[mask] = torch.sigmoid(o[..., 1:]) * (1 + 2 * 0.001) - 0.001
        deform[mask] = dx
    return {"bg_color": color * alpha + bg * (1 - alpha), "torso_alpha": alpha, "deform": deform, "mask": mask}
