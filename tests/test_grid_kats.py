"""Multi-resolution grid encoder (gridencoder.cu:36-165): known answers derived from the GEOMETRY of the encoder -- the published
instant-ngp definition -- and not from either transcription of the kernel (oracle/ernerf_ref.c and csrc/mf_nerf*.hip share an author; the reference's
gridencoder.cu does not build on ROCm 7.2, so this row cannot be pinned to the reference kernel itself: VERDICT r02 "what's missing" item 3).

  * a dense level whose table holds an affine function of the vertex coordinates must return that affine function of the continuous position
    (multilinear interpolation reproduces affine functions exactly) -- for align_corners False (pos = x * scale + 0.5, row stride res + 1) and True
    (pos = x * scale, row stride res);
  * a hashed level: the table entry a vertex reads is `(gx * 1) ^ (gy * 2654435761) [^ (gz * 805459861)]  mod 2^32  mod hashmap_size` -- the primes of
    gridencoder.cu:42, computed here by hand in Python integers -- so a table holding known values at known slots must come back multilinearly
    weighted by the fractional position; at vertex positions (levels whose scale is a power of two, reachable exactly in fp32) it comes back EXACTLY;
  * out-of-range inputs give zeros.

Held on three implementations: the C oracle (CPU), the drop-in extension `_gridencoder.grid_encode_forward` (k_grid_encode) and -- through a field
whose audio-attention net is wired to pass ONE grid feature straight to the `ambient_aud` output -- the gather inside k_nerf_field_fused, the kernel the
render loop actually runs."""
import numpy as np
import pytest
import torch

from test_ernerf import ref, o_grid, grid_offsets, _mods, _cu   # noqa: F401  (ref is a fixture)

PRIMES = (1, 2654435761, 805459861)                               # gridencoder.cu:42


def level_geometry(level, S, H, align):
    """scale and resolution of a level as the encoder defines them (gridencoder.cu:122-124), in fp32 like every implementation must"""
    scale = np.float32(np.exp2(np.float32(level) * np.float32(S))) * np.float32(H) - np.float32(1.0)
    res = int(np.ceil(scale)) + 1
    return float(scale), res, (res if align else res + 1)


def vertex_slot(g, stride, hashmap_size, hashed):
    """table slot of grid vertex g (tuple of ints): row-major with `stride` on a dense level, the prime hash on a hashed one"""
    if hashed:
        h = 0
        for d, gd in enumerate(g):
            h ^= (gd * PRIMES[d]) & 0xFFFFFFFF
        return h % hashmap_size
    idx, st = 0, 1
    for gd in g:
        idx += gd * st
        st *= stride
    return idx % hashmap_size


def expected_level(x01, table, level, S, H, align, hashmap_size, force_tiled=False):
    """float64 value of one level at positions x01 [B, D] from the definition: multilinear weights of the 2^D surrounding vertices x their table entries.
    Also returns, per point, the distance of the position from the nearest cell boundary (points within an fp32 rounding of one are not compared)."""
    B, D = x01.shape
    scale, res, stride = level_geometry(level, S, H, align)
    hashed = (not force_tiled) and stride ** D > hashmap_size
    pos = x01.astype(np.float64) * scale + (0.0 if align else 0.5)
    g0 = np.floor(pos).astype(np.int64)
    f = pos - g0
    out = np.zeros(B)
    for corner in range(1 << D):
        w = np.ones(B)
        g = g0.copy()
        for d in range(D):
            if corner >> d & 1:
                w *= f[:, d]; g[:, d] += 1
            else:
                w *= 1 - f[:, d]
        out += w * np.array([table[vertex_slot(tuple(int(v) for v in gi), stride, hashmap_size, hashed)] for gi in g])
    margin = np.minimum(f, 1 - f).min(axis=1)
    return out, margin, hashed


def build_tables(offs, D, L, S, H, align, gridtype, seed):
    """per level: an affine function of the vertex coordinates on dense levels, slot-dependent pseudo-random values of a few exactly representable
    magnitudes on hashed ones; returns emb [n, 1] and, per level, (table as float64 array)"""
    rng = np.random.default_rng(seed)
    emb = np.zeros((offs[-1], 1), np.float32)
    for l in range(L):
        n = int(offs[l + 1] - offs[l])
        scale, res, stride = level_geometry(l, S, H, align)
        if stride ** D <= n:                                          # a level that fits its table: affine in the vertex coordinates
            a = rng.integers(1, 8, D) / 64.0
            c = rng.integers(1, 8) / 8.0
            idx = np.arange(n)
            coords = np.stack([(idx // stride ** d) % stride for d in range(D)], -1)
            emb[offs[l]:offs[l + 1], 0] = (coords * a).sum(-1) + c
        else:                                                         # hashed, or tiled and wrapped modulo the table: known values at known slots
            emb[offs[l]:offs[l + 1], 0] = (rng.integers(0, 251, n) + 1) / 256.0
    return emb


def check_levels(got_LB, x01, emb, offs, D, L, S, H, align, gridtype, tol):
    """got_LB: [L, B] values of every level from an implementation"""
    n_hashed = n_dense = 0
    for l in range(L):
        table = emb[offs[l]:offs[l + 1], 0].astype(np.float64)
        want, margin, hashed = expected_level(x01, table, l, S, H, align, len(table), force_tiled=gridtype == 1)
        ok = margin > 2e-3                                         # fp32 rounding of x * scale + 0.5 cannot flip the cell of these points
        assert ok.mean() > 0.9
        err = np.abs(got_LB[l][ok] - want[ok]).max()
        assert err <= tol, (l, "hashed" if hashed else "dense", err)
        n_hashed += hashed; n_dense += not hashed
    return n_dense, n_hashed


CASES = [  # D, L, H, per-level scale, log2 hashmap, gridtype (0 hash / 1 tiled), align_corners
    (2, 12, 64, 2 ** (3 / 11), 14, 0, False),      # the head's tri-plane encoders (network.py: 12 levels 64 -> 512, 2^14 entries): levels 4-11 hashed
    (2, 6, 16, 1.5, 10, 0, True),
    (3, 5, 8, 1.6, 12, 0, False),
    (3, 4, 8, 1.5, 12, 0, True),
    (2, 16, 16, 2 ** (7 / 15), 16, 1, False),      # the torso's tiled encoder (network.py:162): every level dense, wrapped modulo the table
]


@pytest.mark.parametrize("D,L,H,pls,log2h,gridtype,align", CASES)
def test_oracle_grid_geometry_kats(ref, D, L, H, pls, log2h, gridtype, align):
    S = float(np.log2(pls))
    offs = grid_offsets(D, L, H, pls, log2h, align)
    emb = build_tables(offs, D, L, S, H, align, gridtype, seed=D * 10 + L)
    x = np.random.default_rng(L).random((400, D)).astype(np.float32) * 0.98 + 0.01
    out = o_grid(ref, x, emb, offs, D, 1, S, H, gridtype, int(align))[:, :, 0]
    nd, nh = check_levels(out, x, emb, offs, D, L, S, H, align, gridtype, tol=2e-4)
    assert nd >= 1 and (gridtype == 1 or nh >= 1)                      # both index forms were exercised (tiled: dense and wrapped levels)
    oob = o_grid(ref, np.array([[1.5] + [0.2] * (D - 1), [0.3] * (D - 1) + [-0.1]], np.float32), emb, offs, D, 1, S, H, gridtype, int(align))
    assert (oob == 0).all()                                            # gridencoder.cu:98-116


def _vertex_case(D, align):
    """one hashed level whose scale is a power of two: a vertex position is exact in fp32 and the level must return the table entry itself"""
    H = 1025 if D == 2 else 65                                        # scale = H - 1 = 1024 / 64 at level 0
    log2h = 14
    offs = grid_offsets(D, 1, H, 2.0, log2h, align)
    n = int(offs[1])
    emb = (np.arange(n, dtype=np.float32) + 1).reshape(n, 1)          # entry t holds t + 1: the value IS the slot
    scale, res, stride = level_geometry(0, 1.0, H, align)
    assert scale == H - 1 and stride ** D > n
    rng = np.random.default_rng(D)
    g = rng.integers(1, res - 1, (300, D))
    x = ((g - (0.0 if align else 0.5)) / scale).astype(np.float32)
    assert np.array_equal((x.astype(np.float64) * scale + (0.0 if align else 0.5)), g)       # exact
    want = np.array([vertex_slot(tuple(int(v) for v in gi), stride, n, True) + 1 for gi in g], np.float32)
    return offs, emb, x, want, H


@pytest.mark.parametrize("D,align", [(2, False), (2, True), (3, False), (3, True)])
def test_oracle_grid_hashed_vertices_exact(ref, D, align):
    offs, emb, x, want, H = _vertex_case(D, align)
    out = o_grid(ref, x, emb, offs, D, 1, 1.0, H, 0, int(align))[0, :, 0]
    np.testing.assert_array_equal(out, want)


@pytest.mark.gpu
@pytest.mark.parametrize("D,L,H,pls,log2h,gridtype,align", CASES)
def test_hip_grid_encoder_geometry_kats(lib_built, D, L, H, pls, log2h, gridtype, align):
    ge = _mods()[1]
    S = float(np.log2(pls))
    offs = grid_offsets(D, L, H, pls, log2h, align)
    emb = build_tables(offs, D, L, S, H, align, gridtype, seed=D * 10 + L)
    x = np.random.default_rng(L + 50).random((2000, D)).astype(np.float32) * 0.98 + 0.01
    out = torch.empty(L, x.shape[0], 1, device="cuda")
    ge.grid_encode_forward(_cu(x), _cu(emb), _cu(offs), out, x.shape[0], D, 1, L, S, H, None, gridtype, align)
    check_levels(out.cpu().numpy()[:, :, 0], x, emb, offs, D, L, S, H, align, gridtype, tol=2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("D,align", [(2, False), (2, True), (3, False), (3, True)])
def test_hip_grid_encoder_hashed_vertices_exact(lib_built, D, align):
    ge = _mods()[1]
    offs, emb, x, want, H = _vertex_case(D, align)
    out = torch.empty(1, x.shape[0], 1, device="cuda")
    ge.grid_encode_forward(_cu(x), _cu(emb), _cu(offs), out, x.shape[0], D, 1, 1, 1.0, H, None, 0, align)
    np.testing.assert_array_equal(out.cpu().numpy()[0, :, 0], want)


@pytest.mark.gpu
def test_hip_fused_field_grid_gather_geometry_kats(lib_built):
    """The gather INSIDE k_nerf_field_fused (its own index arithmetic: one multiply per index form, a mask for the power-of-two hash tables) against the
    same geometry: `aud_ch_att_net` (36 -> 64 -> 32, bias-free, ReLU between: network.py:79-90) is wired as hidden_0 = feature_k, out_0 = hidden_0, so
    that `ambient_aud = ||aud_ch_att||` (network.py:268) IS grid feature k (positive tables: the ReLU is the identity) for the plane / level k names."""
    from mere_fusion_amd import weights as W
    from mere_fusion_amd.ernerf.field import HipNeRFField, grid_geometry
    offs, pls = grid_geometry()
    S, H, L = float(np.log2(pls)), 64, 12
    sd = W.make_ernerf_field_state_dict(int(offs[-1]), 0)
    tables = {}
    for p, plane in enumerate(("xy", "yz", "xz")):
        tables[plane] = build_tables(offs, 2, L, S, H, False, 0, seed=100 + p)
        sd[f"encoder_{plane}.embeddings"] = torch.from_numpy(tables[plane])
    rng = np.random.default_rng(7)
    M = 4096
    x = (rng.random((M, 3)).astype(np.float32) * 1.9 - 0.95)            # inside the bound-1 box
    d = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (M, 1))
    planes = {"xy": x[:, :2], "yz": x[:, 1:], "xz": np.stack([x[:, 0], x[:, 2]], -1)}                     # split_xyz, network.py:204-208
    g = torch.Generator().manual_seed(0)
    enc_a, ind, eye = torch.randn(1, 32, generator=g), torch.randn(1, 4, generator=g) * 0.1, torch.tensor([[0.4]])
    seen = {False: 0, True: 0}
    for p, plane in enumerate(("xy", "yz", "xz")):
        x01 = ((planes[plane] + np.float32(1.0)) / np.float32(2.0)).astype(np.float32)                     # grid.py:144
        for l in (0, 2, 3, 4, 7, 11):                                  # dense levels 0-3, hashed 4-11
            k = p * L + l                                              # enc_x = [xy levels | yz levels | xz levels], network.py:211-219
            w0, w1 = torch.zeros(64, 36), torch.zeros(32, 64)
            w0[0, k] = 1.0
            w1[0, 0] = 1.0
            sdk = dict(sd)
            sdk["aud_ch_att_net.net.0.weight"], sdk["aud_ch_att_net.net.1.weight"] = w0, w1
            field = HipNeRFField(sdk, max_samples=M)
            amb = field.forward(torch.from_numpy(x).cuda(), torch.from_numpy(d).cuda(), enc_a.cuda(), ind.cuda(), eye.cuda())[2].cpu().numpy()[:, 0]
            table = tables[plane][offs[l]:offs[l + 1], 0].astype(np.float64)
            want, margin, hashed = expected_level(x01, table, l, S, H, False, len(table))
            ok = margin > 2e-3
            err = np.abs(amb[ok] - want[ok]).max()
            # two bf16x3 Linears carry the feature to the output: ~2^-16 relative on values <= 2
            assert ok.mean() > 0.9 and err <= 3e-4, (plane, l, "hashed" if hashed else "dense", err)
            seen[hashed] += 1
            del field
    assert seen[False] >= 9 and seen[True] >= 9
