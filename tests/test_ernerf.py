"""ER-NeRF inference kernels (SURVEY 8a rows a16-a19, a21, a22's frequency encoder).

CPU: the plain-C oracle (oracle/ernerf_ref.c, PARITY UNPINNED -- the CUDA reference cannot run here) against hand-derived
known answers.  GPU: the HIP kernels, called through the extension-module shims the reference's wrappers import
(`_raymarching_face`, `_gridencoder`, `_shencoder`, `_freqencoder`), against the oracle on the same seeded inputs:
bit-exact wherever only +,-,*,/ and exact libm calls are involved (both sides are built without FMA contraction), a written
tolerance where expf / sinf are."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
FLT_MAX = np.finfo(np.float32).max


@pytest.fixture(scope="module")
def ref():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "libernerfref.so"))
    lib.ref_morton3d.restype = C.c_uint32
    lib.ref_morton3d.argtypes = [C.c_uint32] * 3
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


F = lambda *s: np.zeros(s, np.float32)


# ---- oracle wrappers (numpy in, numpy out) ----------------------------------------------------------------------------
def o_near_far(ref, ro, rd, aabb, min_near):
    n = ro.shape[0]
    nears, fars = F(n), F(n)
    ref.ref_near_far_from_aabb(_p(ro), _p(rd), _p(aabb), C.c_uint32(n), C.c_float(min_near), _p(nears), _p(fars))
    return nears, fars


def o_march(ref, n_step, alive, rays_t, ro, rd, bound, dt_gamma, max_steps, cascades, H, grid, nears, fars, noises):
    na = alive.shape[0]
    xyzs, dirs, deltas = F(na * n_step, 3), F(na * n_step, 3), F(na * n_step, 2)
    ref.ref_march_rays(C.c_uint32(na), C.c_uint32(n_step), _p(alive), _p(rays_t), _p(ro), _p(rd), C.c_float(bound), C.c_float(dt_gamma),
                       C.c_uint32(max_steps), C.c_uint32(cascades), C.c_uint32(H), _p(grid), _p(nears), _p(fars), _p(xyzs), _p(dirs),
                       _p(deltas), _p(noises))
    return xyzs, dirs, deltas


def o_composite(ref, n_step, T_thresh, alive, rays_t, sig, rgb, deltas, aa, ae, unc, acc):
    alive, rays_t = alive.copy(), rays_t.copy()
    acc = {k: v.copy() for k, v in acc.items()}
    ref.ref_composite_rays_triplane(C.c_uint32(alive.shape[0]), C.c_uint32(n_step), C.c_float(T_thresh), _p(alive), _p(rays_t), _p(sig),
                                    _p(rgb), _p(deltas), _p(aa), _p(ae), _p(unc), _p(acc["weights_sum"]), _p(acc["depth"]), _p(acc["image"]),
                                    _p(acc["amb_aud_sum"]), _p(acc["amb_eye_sum"]), _p(acc["uncertainty_sum"]))
    return alive, rays_t, acc


def o_grid(ref, x, emb, offsets, D, Cc, S, H, gridtype, align):
    B, L = x.shape[0], offsets.shape[0] - 1
    out = F(L, B, Cc)
    ref.ref_grid_encode_forward(_p(x), _p(emb), _p(offsets), _p(out), C.c_uint32(B), C.c_uint32(D), C.c_uint32(Cc), C.c_uint32(L),
                                C.c_float(S), C.c_uint32(H), C.c_uint32(gridtype), C.c_int(align))
    return out


def o_sh(ref, d, degree):
    out = F(d.shape[0], degree * degree)
    ref.ref_sh_encode_forward(_p(d), _p(out), C.c_uint32(d.shape[0]), C.c_uint32(degree))
    return out


def o_freq(ref, x, deg):
    B, D = x.shape
    Cc = D + 2 * D * deg
    out = F(B, Cc)
    ref.ref_freq_encode_forward(_p(x), C.c_uint32(B), C.c_uint32(D), C.c_uint32(deg), C.c_uint32(Cc), _p(out))
    return out


def grid_offsets(D, L, H, per_level_scale, log2_hashmap, align=False):
    """GridEncoder.__init__, grid.py:108-123."""
    offs, off = [], 0
    for i in range(L):
        res = int(np.ceil(H * per_level_scale ** i))
        n = min(2 ** log2_hashmap, (res if align else res + 1) ** D)
        n = int(np.ceil(n / 8) * 8)
        offs.append(off)
        off += n
    offs.append(off)
    return np.array(offs, np.int32)


# ---- CPU known-answer tests ------------------------------------------------------------------------------------------------
AABB = np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32)      # aabb_infer at bound 1, renderer.py:86-89


def test_oracle_near_far_kats(ref):
    ro = np.array([[-2, 0, 0], [-2, 0, 0], [0, 0, 0], [0, 2, 0]], np.float32)
    rd = np.array([[1, 1e-9, 1e-9], [0, 1, 1e-9], [1e-9, 1e-9, 1], [1, 1e-9, 1e-9]], np.float32)
    nears, fars = o_near_far(ref, ro, rd, AABB, 0.05)
    assert nears[0] == 1.0 and fars[0] == 3.0                                  # enters x = -1 at t = 1, leaves x = 1 at t = 3
    assert nears[1] == FLT_MAX and fars[1] == FLT_MAX                          # parallel to the box, outside its x slab
    assert nears[2] == np.float32(0.05) and fars[2] == 1.0                     # origin inside: near clamps to min_near
    assert nears[3] == FLT_MAX                                                 # above the y slab, moving along x


def test_oracle_morton_kats(ref):
    assert [ref.ref_morton3d(1, 0, 0), ref.ref_morton3d(0, 1, 0), ref.ref_morton3d(0, 0, 1)] == [1, 2, 4]
    assert ref.ref_morton3d(3, 3, 3) == 63 and ref.ref_morton3d(127, 127, 127) == 2 ** 21 - 1
    assert ref.ref_morton3d(5, 0, 0) == 0b1000001                             # x bits land on positions 0, 3, 6, ...


def test_oracle_sh_kats(ref):
    out = o_sh(ref, np.array([[0, 0, 1], [1, 0, 0]], np.float32), 4)
    c = 0.5 * np.sqrt(1 / np.pi)
    np.testing.assert_allclose(out[:, 0], c, rtol=1e-7)
    np.testing.assert_allclose(out[0, 2], np.sqrt(3 / (4 * np.pi)), rtol=1e-6)               # Y_1^0 at the pole
    np.testing.assert_allclose(out[0, 6], 0.25 * np.sqrt(5 / np.pi) * 2, rtol=1e-6)          # Y_2^0 = sqrt(5/pi)/4 (3z^2-1)
    np.testing.assert_allclose(out[0, 12], 0.25 * np.sqrt(7 / np.pi) * 2, rtol=1e-6)         # Y_3^0 = sqrt(7/pi)/4 (5z^3-3z)
    np.testing.assert_allclose(out[1, 3], -np.sqrt(3 / (4 * np.pi)), rtol=1e-6)
    assert out[0, 1] == 0 and out[0, 3] == 0 and out[0, 4] == 0


def test_oracle_freq_kats(ref):
    x = np.array([[0.25, -0.5]], np.float32)
    out = o_freq(ref, x, 3)
    assert out.shape == (1, 14)
    want = [0.25, -0.5]
    for f in range(3):
        for ph in (0, np.pi / 2):                                              # freq.py: [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...]
            want += [np.sin(0.25 * 2 ** f + ph), np.sin(-0.5 * 2 ** f + ph)]
    np.testing.assert_allclose(out[0], np.array(want, np.float32), atol=1e-6)


def test_oracle_grid_reproduces_affine_tables(ref):
    # dense (tiled) levels holding an affine function of the vertex index: bilinear interpolation must return the same affine
    # function of the continuous position x * scale + 0.5 (gridencoder.cu:131-137)
    D, L, H, pls = 2, 3, 8, 2.0
    offs = grid_offsets(D, L, H, pls, 19)
    emb = np.zeros((offs[-1], 1), np.float32)
    for l in range(L):
        res = int(np.ceil(np.exp2(l * np.log2(pls)) * H - 1.0)) + 1
        for j in range(res + 1):
            for i in range(res + 1):
                idx = i + j * (res + 1)
                if idx < offs[l + 1] - offs[l]:
                    emb[offs[l] + idx, 0] = 0.5 * i - 0.25 * j + 3
    x = np.random.default_rng(0).random((64, 2), np.float32) * 0.98
    out = o_grid(ref, x, emb, offs, D, 1, float(np.log2(pls)), H, 1, 0)
    for l in range(L):
        scale = np.float32(np.exp2(np.float32(l) * np.float32(np.log2(pls))) * H - 1.0)
        pos = x * scale + 0.5
        np.testing.assert_allclose(out[l, :, 0], 0.5 * pos[:, 0] - 0.25 * pos[:, 1] + 3, rtol=2e-6)
    oob = o_grid(ref, np.array([[1.5, 0.2], [0.3, -0.1]], np.float32), emb, offs, D, 1, 1.0, H, 1, 0)
    assert (oob == 0).all()                                                    # gridencoder.cu:98-116


def test_oracle_composite_closed_form(ref):
    n_step = 3
    alive = np.array([0, 1], np.int32)
    rays_t = np.array([0.1, 0.2], np.float32)
    sig = np.array([[2.0, 0, 0], [1.0, 3.0, 0.5]], np.float32)
    deltas = np.array([[[0.5, 0.6], [0, 0], [0, 0]], [[0.1, 0.3], [0.1, 0.4], [0.1, 0.5]]], np.float32)
    rgb = np.full((2, n_step, 3), 0.5, np.float32)
    one = np.ones((2, n_step), np.float32)
    acc = dict(weights_sum=F(2), depth=F(2), image=F(2, 3), amb_aud_sum=F(2), amb_eye_sum=F(2), uncertainty_sum=F(2))
    a2, t2, acc2 = o_composite(ref, n_step, 1e-4, alive, rays_t, sig, rgb, deltas, one, 2 * one, one, acc)
    a0 = 1 - np.exp(-1.0)
    np.testing.assert_allclose(acc2["weights_sum"][0], a0, rtol=1e-6)
    np.testing.assert_allclose(acc2["image"][0], 0.5 * a0, rtol=1e-6)
    np.testing.assert_allclose(acc2["depth"][0], a0 * 0.6, rtol=1e-6)
    assert a2[0] == -1 and t2[0] == np.float32(0.1)                            # dt == 0 at step 1: terminated, rays_t untouched
    al = 1 - np.exp(-np.array([0.1, 0.3, 0.05]))
    w = np.array([al[0], al[1] * (1 - al[0]), al[2] * (1 - al[0] - al[1] * (1 - al[0]))])
    np.testing.assert_allclose(acc2["weights_sum"][1], w.sum(), rtol=1e-6)
    np.testing.assert_allclose(acc2["depth"][1], (w * [0.3, 0.4, 0.5]).sum(), rtol=1e-6)
    assert a2[1] == 1 and t2[1] == np.float32(0.5)                             # survived all steps: rays_t = last t
    assert acc2["amb_aud_sum"][1] == 3 and acc2["amb_eye_sum"][1] == 6          # ambient terms are plain sums (raymarching.cu:2207-2208)


# ---- known answers for march_rays derived from the DDA GEOMETRY, not from either transcription of the kernel -----------------------
def _morton_np(x, y, z):
    def expand(v):
        v = v.astype(np.uint64)
        v = (v | (v << 16)) & 0xFF0000FF
        v = (v | (v << 8)) & 0x0F00F00F
        v = (v | (v << 4)) & 0xC30C30C3
        v = (v | (v << 2)) & 0x49249249
        return v
    return (expand(x) | (expand(y) << 1) | (expand(z) << 2)).astype(np.int64)


def _bitfield(H, cascades, occupied):
    """occupied(level, nx, ny, nz) -> bool array over the full index grid; Morton order inside each cascade (raymarching.cu:894-895)."""
    bits = np.zeros(cascades * H ** 3, np.uint8)
    n = np.arange(H)
    nx, ny, nz = np.meshgrid(n, n, n, indexing="ij")
    for lvl in range(cascades):
        idx = lvl * H ** 3 + _morton_np(nx.ravel(), ny.ravel(), nz.ravel())
        bits[idx] = occupied(lvl, nx.ravel(), ny.ravel(), nz.ravel())
    return np.packbits(bits.reshape(-1, 8), axis=1, bitorder="little").reshape(-1)


def dda_case_slab():
    """One cascade, bound 1, H = 128: a ray along +x enters the box at x = -1 (t = 1).  With dt_gamma = 1/256 and max_steps = 16 the step
    is pinned to dt_max = dt_min = 2 sqrt(3) / 128 = 0.02706 > one voxel (1/64), so in empty space every DDA skip advances by exactly one
    step: t_k = 1 + k * dt.  Only the voxel slab nx in [96, 100) -- x in [0.5, 0.5625) -- is occupied: the ray's positions x_k = -1 + k dt
    fall inside it for k = 56, 57 only (56 dt = 1.5155, 57 dt = 1.5426, 58 dt = 1.5697 > 1.5625).  Expected: exactly two samples."""
    H = 128
    grid = _bitfield(H, 1, lambda lvl, nx, ny, nz: ((nx >= 96) & (nx < 100)).astype(np.uint8))
    ro = np.array([[-2.0, 0.1, 0.1]], np.float32)
    rd = np.array([[1.0, 1e-9, 1e-9]], np.float32)
    dt = np.float32(2 * np.sqrt(np.float32(3.0)) / 128)
    t = np.float32(1.0)
    ts = []
    for k in range(60):                                   # the algorithm's own accumulation: t += dt in fp32
        if k in (56, 57):
            ts.append(t)
        t = np.float32(t + dt)
    want_x = np.array([np.float32(-2.0) + tk for tk in ts], np.float32)
    return dict(H=H, grid=grid, ro=ro, rd=rd, bound=1.0, cascades=1, max_steps=16, dt_gamma=1 / 256, n_step=8, dt=dt, want_t=np.array(ts, np.float32),
                want_x=want_x)


def dda_case_cascade_switch():
    """Two cascades, bound 2, H = 128, dt_gamma = 0, max_steps = 1024: dt = dt_min = 2 sqrt(3) / 1024 = 0.003383, small enough that the
    mip level comes from the POSITION alone (mip_from_dt: dt * H / 2 = 0.2165 < 0.5 -> level 0).  Cascade 1 (|x| in [1, 2)) is fully
    occupied, cascade 0 (|x| < 1) is empty, so a ray along +x must emit samples exactly while |x| >= 1: for t in [1, 2) and [4, 5) of
    o = (-3, .1, .1), none in between -- the switch sits at |x| = 1 to within one step."""
    H = 128
    grid = _bitfield(H, 2, lambda lvl, nx, ny, nz: np.full(nx.shape, lvl == 1, np.uint8))
    ro = np.array([[-3.0, 0.1, 0.1]], np.float32)
    rd = np.array([[1.0, 1e-9, 1e-9]], np.float32)
    return dict(H=H, grid=grid, ro=ro, rd=rd, bound=2.0, cascades=2, max_steps=1024, dt_gamma=0.0, n_step=700, dt=np.float32(2 * np.sqrt(3.0) / 1024))


def _check_slab(xyzs, dirs, deltas, c):
    d = deltas.reshape(-1, 2)
    n = int((d[:, 0] > 0).sum())
    assert n == 2, n                                                            # exactly the two positions inside the occupied slab
    np.testing.assert_allclose(d[:2, 0], c["dt"], rtol=1e-6)
    np.testing.assert_allclose(d[:2, 1], c["want_t"] + c["dt"], rtol=1e-6)       # deltas[1] = t after the step (depth)
    np.testing.assert_allclose(xyzs.reshape(-1, 3)[:2, 0], c["want_x"], atol=2e-6)
    np.testing.assert_allclose(xyzs.reshape(-1, 3)[:2, 1:], 0.1, atol=1e-6)
    assert (xyzs.reshape(-1, 3)[2:] == 0).all() and (d[2:] == 0).all()           # the rest of the slot stays zero-filled (raymarching.py:383-385)
    np.testing.assert_array_equal(dirs.reshape(-1, 3)[:2], np.repeat(c["rd"], 2, axis=0))


def _check_cascade_switch(xyzs, deltas, c):
    d = deltas.reshape(-1, 2)
    live = d[:, 0] > 0
    x = xyzs.reshape(-1, 3)[live, 0]
    dt = float(c["dt"])
    assert np.all(np.abs(x) >= 1.0 - 1e-6) and np.all(np.abs(x) < 2.0 + 1e-6)     # never a sample from the empty inner cascade
    left, right = x[x < 0], x[x > 0]
    # each outer segment is 1 long: 1 / dt = 295.6 steps; the DDA re-entry can cost at most a step at either end
    assert abs(len(left) - 1.0 / dt) <= 2 and abs(len(right) - 1.0 / dt) <= 2, (len(left), len(right))
    assert left.max() < -1.0 + 1e-6 and left.max() > -1.0 - 1.5 * dt               # last sample before the switch at x = -1
    assert right.min() >= 1.0 - 1e-6 and right.min() < 1.0 + 1.5 * dt              # first sample after the switch at x = +1
    np.testing.assert_allclose(np.diff(left), dt, rtol=2e-3)
    np.testing.assert_allclose(np.diff(right), dt, rtol=2e-3)


def test_oracle_march_dda_geometry_kats(ref):
    c = dda_case_slab()
    aabb = np.array([-1, -0.5, -1, 1, 0.5, 1], np.float32)
    nears, fars = o_near_far(ref, c["ro"], c["rd"], aabb, 0.05)
    assert nears[0] == 1.0 and fars[0] == 3.0
    out = o_march(ref, c["n_step"], np.zeros(1, np.int32), nears, c["ro"], c["rd"], c["bound"], c["dt_gamma"], c["max_steps"], c["cascades"], c["H"],
                  c["grid"], nears, fars, np.zeros(1, np.float32))
    _check_slab(*out, c)
    c = dda_case_cascade_switch()
    aabb = np.array([-2, -1, -2, 2, 1, 2], np.float32)
    nears, fars = o_near_far(ref, c["ro"], c["rd"], aabb, 0.05)
    assert nears[0] == 1.0 and fars[0] == 5.0
    xyzs, dirs, deltas = o_march(ref, c["n_step"], np.zeros(1, np.int32), nears, c["ro"], c["rd"], c["bound"], c["dt_gamma"], c["max_steps"],
                                 c["cascades"], c["H"], c["grid"], nears, fars, np.zeros(1, np.float32))
    _check_cascade_switch(xyzs, deltas, c)


@pytest.mark.gpu
def test_hip_march_dda_geometry_kats(lib_built):
    """The same geometry-derived expectations on the HIP kernel directly (no oracle in the loop: breaks the common mode of two transcriptions)."""
    from mere_fusion_amd.ernerf import _raymarching_face as rm
    for c, aabb, check in ((dda_case_slab(), [-1, -0.5, -1, 1, 0.5, 1], "slab"), (dda_case_cascade_switch(), [-2, -1, -2, 2, 1, 2], "switch")):
        ro, rd = _cu(c["ro"]), _cu(c["rd"])
        nears, fars = torch.empty(1, device="cuda"), torch.empty(1, device="cuda")
        rm.near_far_from_aabb(ro, rd, _cu(np.array(aabb, np.float32)), 1, 0.05, nears, fars)
        n_step = c["n_step"]
        xyzs, dirs, deltas = (torch.zeros(n_step, k, device="cuda") for k in (3, 3, 2))
        rm.march_rays(1, n_step, _cu(np.zeros(1, np.int32)), nears.clone(), ro, rd, c["bound"], c["dt_gamma"], c["max_steps"], c["cascades"], c["H"],
                      _cu(c["grid"]), nears, fars, xyzs, dirs, deltas, torch.zeros(1, device="cuda"))
        if check == "slab":
            _check_slab(xyzs.cpu().numpy(), dirs.cpu().numpy(), deltas.cpu().numpy(), c)
        else:
            _check_cascade_switch(xyzs.cpu().numpy(), deltas.cpu().numpy(), c)


def _scene(n_rays, seed, H=128, cascades=1, density=0.3):
    rng = np.random.default_rng(seed)
    ro = (rng.standard_normal((n_rays, 3)) * 0.1 + [0, 0, -2.2]).astype(np.float32)
    target = rng.uniform(-0.9, 0.9, (n_rays, 3)).astype(np.float32) * [1, 0.5, 1]
    rd = target - ro
    rd = (rd / np.linalg.norm(rd, axis=1, keepdims=True)).astype(np.float32)
    grid = (rng.random(cascades * H ** 3 // 8) < density).astype(np.uint8) * rng.integers(1, 256, cascades * H ** 3 // 8).astype(np.uint8)
    return ro, rd, grid


def test_oracle_march_empty_and_full_grid(ref):
    ro, rd, _ = _scene(64, 1)
    nears, fars = o_near_far(ref, ro, rd, AABB, 0.05)
    alive = np.arange(64, dtype=np.int32)
    noises = np.zeros(64, np.float32)
    empty = np.zeros(128 ** 3 // 8, np.uint8)
    xyzs, dirs, deltas = o_march(ref, 4, alive, nears, ro, rd, 1.0, 1 / 256, 16, 1, 128, empty, nears, fars, noises)
    assert (deltas == 0).all() and (xyzs == 0).all()                            # nothing occupied: no samples
    full = np.full(128 ** 3 // 8, 255, np.uint8)
    xyzs, dirs, deltas = o_march(ref, 4, alive, nears, ro, rd, 1.0, 1 / 256, 16, 1, 128, full, nears, fars, noises)
    hit = fars > nears + 0.5
    d = deltas.reshape(64, 4, 2)
    dt_max = np.float32(2 * np.sqrt(3) / 128)
    assert (d[hit, :, 0] > 0).all() and (d[hit, :, 0] <= dt_max * 1.0001).all()
    np.testing.assert_allclose(d[hit, 0, 1], nears[hit] + d[hit, 0, 0], rtol=1e-6)   # first sample sits at t = near
    p = xyzs.reshape(64, 4, 3)[hit, 0]
    np.testing.assert_allclose(p, np.clip(ro[hit] + nears[hit, None] * rd[hit], -1, 1), atol=1e-6)
    np.testing.assert_array_equal(dirs.reshape(64, 4, 3)[hit, 2], rd[hit])


# ---- GPU parity ----------------------------------------------------------------------------------------------------------
def _mods():
    sys.path.insert(0, os.path.join(ROOT, "mere-fusion_amd", "dropin"))
    import _freqencoder, _gridencoder, _raymarching_face, _shencoder
    return _raymarching_face, _gridencoder, _shencoder, _freqencoder


def _cu(a):
    return torch.from_numpy(a).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("n_rays,cascades,n_step", [(4096, 1, 1), (4096, 1, 8), (1000, 2, 3), (262144, 1, 2)])
def test_hip_near_far_and_march_bit_exact(lib_built, ref, n_rays, cascades, n_step):
    rm = _mods()[0]
    ro, rd, grid = _scene(n_rays, n_rays + cascades, cascades=cascades)
    bound = float(2 ** (cascades - 1))
    aabb = AABB * bound
    nears_o, fars_o = o_near_far(ref, ro, rd, aabb, 0.05)
    d_ro, d_rd = _cu(ro), _cu(rd)
    nears, fars = torch.empty(n_rays, device="cuda"), torch.empty(n_rays, device="cuda")
    rm.near_far_from_aabb(d_ro, d_rd, _cu(aabb), n_rays, 0.05, nears, fars)
    np.testing.assert_array_equal(nears.cpu().numpy(), nears_o)
    np.testing.assert_array_equal(fars.cpu().numpy(), fars_o)
    rng = np.random.default_rng(7)
    alive = rng.permutation(n_rays).astype(np.int32)[: max(1, n_rays * 3 // 4)]
    noises = rng.random(alive.shape[0], np.float32)
    rays_t = nears_o.copy()
    want = o_march(ref, n_step, alive, rays_t, ro, rd, bound, 1 / 256, 16, cascades, 128, grid, nears_o, fars_o, noises)
    na = alive.shape[0]
    xyzs, dirs, deltas = (torch.zeros(na * n_step, k, device="cuda") for k in (3, 3, 2))
    rm.march_rays(na, n_step, _cu(alive), _cu(rays_t), d_ro, d_rd, bound, 1 / 256, 16, cascades, 128, _cu(grid), nears, fars, xyzs, dirs,
                  deltas, _cu(noises))
    for got, w, name in zip((xyzs, dirs, deltas), want, ("xyzs", "dirs", "deltas")):
        np.testing.assert_array_equal(got.cpu().numpy(), w, err_msg=name)
    assert (want[2][:, 0] > 0).mean() > 0.05                                   # the scene does produce samples


@pytest.mark.gpu
@pytest.mark.parametrize("n_alive,n_step", [(5000, 1), (5000, 4), (100000, 8)])
def test_hip_composite_matches_oracle(lib_built, ref, n_alive, n_step):
    rm = _mods()[0]
    rng = np.random.default_rng(n_alive + n_step)
    N = n_alive * 2
    alive = rng.permutation(N).astype(np.int32)[:n_alive]
    rays_t = rng.random(N, np.float32)
    sig = (rng.random((n_alive, n_step), np.float32) * 40).astype(np.float32)
    rgb = rng.random((n_alive, n_step, 3), np.float32)
    dt = (rng.random((n_alive, n_step), np.float32) * 0.03 + 0.001).astype(np.float32)
    dt[rng.random((n_alive, n_step)) < 0.1] = 0                                # "no sample" markers
    deltas = np.stack([dt, np.cumsum(dt, 1) + 0.2], -1).astype(np.float32)
    aa, ae, unc = (rng.random((n_alive, n_step), np.float32) for _ in range(3))
    acc = dict(weights_sum=(rng.random(N, np.float32) * 0.999).astype(np.float32), depth=rng.random(N, np.float32), image=rng.random((N, 3), np.float32),
               amb_aud_sum=rng.random(N, np.float32), amb_eye_sum=rng.random(N, np.float32), uncertainty_sum=rng.random(N, np.float32))
    a_o, t_o, acc_o = o_composite(ref, n_step, 1e-4, alive, rays_t, sig, rgb, deltas, aa, ae, unc, acc)
    d_alive, d_t = _cu(alive.copy()), _cu(rays_t.copy())
    d_acc = {k: _cu(v.copy()) for k, v in acc.items()}
    rm.composite_rays_triplane(n_alive, n_step, 1e-4, d_alive, d_t, _cu(sig), _cu(rgb), _cu(deltas), _cu(aa), _cu(ae), _cu(unc), d_acc["weights_sum"],
                               d_acc["depth"], d_acc["image"], d_acc["amb_aud_sum"], d_acc["amb_eye_sum"], d_acc["uncertainty_sum"])
    # __expf vs libm expf: a couple of ulp per alpha, accumulated over <= 8 steps; T_thresh decisions are far from the noise
    np.testing.assert_array_equal(d_alive.cpu().numpy(), a_o)
    np.testing.assert_array_equal(d_t.cpu().numpy(), t_o)
    for k in acc:
        np.testing.assert_allclose(d_acc[k].cpu().numpy(), acc_o[k], rtol=2e-6, atol=2e-6, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("D,Cc,L,H,pls,log2h,gridtype,align", [(2, 1, 12, 64, 1.31951, 14, 0, False), (2, 2, 16, 16, 1.38, 17, 1, False),
                                                                (3, 2, 8, 16, 1.5, 12, 0, False), (2, 4, 4, 8, 2.0, 10, 0, True), (3, 8, 3, 4, 2.0, 19, 1, False)])
def test_hip_grid_encoder_bit_exact(lib_built, ref, D, Cc, L, H, pls, log2h, gridtype, align):
    ge = _mods()[1]
    rng = np.random.default_rng(D * 100 + Cc)
    offs = grid_offsets(D, L, H, pls, log2h, align)
    emb = rng.uniform(-1e-1, 1e-1, (offs[-1], Cc)).astype(np.float32)
    B = 20000
    x = rng.random((B, D), np.float32)
    x[:7] = [[0.0] * D, [1.0] * D, [1.0000001] * D, [-1e-7] * D, [0.5] * D, [0.999999] * D, [1e-7] * D]   # edges and out-of-range rows
    S = float(np.log2(pls))
    want = o_grid(ref, x, emb, offs, D, Cc, S, H, gridtype, int(align))
    out = torch.empty(L, B, Cc, device="cuda")
    ge.grid_encode_forward(_cu(x), _cu(emb), _cu(offs), out, B, D, Cc, L, S, H, None, gridtype, align)
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    out2 = torch.empty(B, L * Cc, device="cuda")
    ge.grid_encode_forward_blc(_cu(x), _cu(emb), _cu(offs), out2, B, D, Cc, L, S, H, gridtype, align)
    np.testing.assert_array_equal(out2.cpu().numpy(), want.transpose(1, 0, 2).reshape(B, L * Cc))      # grid.py:52


@pytest.mark.gpu
def test_hip_sh_and_freq_encoders(lib_built, ref):
    _, _, sh, fq = _mods()
    rng = np.random.default_rng(3)
    d = rng.standard_normal((50000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for degree in (1, 2, 3, 4):
        out = torch.empty(d.shape[0], degree * degree, device="cuda")
        sh.sh_encode_forward(_cu(d), out, d.shape[0], 3, degree, None)
        np.testing.assert_array_equal(out.cpu().numpy(), o_sh(ref, d, degree))
    for D, deg in ((2, 8), (6, 3), (3, 10)):                                    # forward_torso uses freq(2, 8) and freq(6, 3), network.py
        x = rng.uniform(-1, 1, (4000, D)).astype(np.float32)
        Cc = D + 2 * D * deg
        out = torch.empty(4000, Cc, device="cuda")
        fq.freq_encode_forward(_cu(x), 4000, D, deg, Cc, out)
        want = o_freq(ref, x, deg)
        # __sinf: absolute error ~ 2^-21 * |argument| (arguments reach 2^9); the identity columns are exact
        np.testing.assert_array_equal(out.cpu().numpy()[:, :D], want[:, :D])
        assert np.abs(out.cpu().numpy() - want).max() <= 5e-4 * (2.0 ** deg / 512) + 2e-6


@pytest.mark.gpu
def test_hip_shims_reject_cpu_tensors_and_training_calls(lib_built):
    rm, ge, sh, fq = _mods()
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        rm.near_far_from_aabb(torch.zeros(4, 3), torch.zeros(4, 3), torch.zeros(6), 4, 0.05, torch.zeros(4), torch.zeros(4))
    with pytest.raises(RuntimeError, match="training"):
        rm.march_rays_train()
    with pytest.raises(RuntimeError, match="training"):
        ge.grid_encode_backward()
    x = torch.zeros(4, 3, device="cuda")
    with pytest.raises(RuntimeError, match="degree"):
        sh.sh_encode_forward(x, torch.zeros(4, 25, device="cuda"), 4, 3, 5, None)


@pytest.mark.gpu
def test_hip_render_loop_properties_full_size(lib_built):
    """cfg 5 size (512 x 512 rays): march -> (synthetic sigma / rgb) -> composite until no ray is alive, renderer.py:246-270.
    Size-independent invariants: weights in [0, 1], every ray ends terminated, depth within [near, far], image within [0, 1]."""
    rm = _mods()[0]
    N = 512 * 512
    ro, rd, grid = _scene(N, 5, density=0.5)
    d_ro, d_rd, d_grid = _cu(ro), _cu(rd), _cu(grid)
    nears, fars = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    rm.near_far_from_aabb(d_ro, d_rd, _cu(AABB), N, 0.05, nears, fars)
    acc = {k: torch.zeros(N, device="cuda") for k in ("weights_sum", "depth", "amb_aud_sum", "amb_eye_sum", "uncertainty_sum")}
    image = torch.zeros(N, 3, device="cuda")
    alive = torch.arange(N, dtype=torch.int32, device="cuda")
    rays_t = nears.clone()
    step, max_steps = 0, 16
    while step < max_steps:
        if step > 0:
            alive = alive[alive >= 0].contiguous()
        n_alive = alive.shape[0]
        if n_alive == 0:
            break
        n_step = max(min(N // n_alive, 8), 1)
        xyzs, dirs, deltas = (torch.zeros(n_alive * n_step, k, device="cuda") for k in (3, 3, 2))
        rm.march_rays(n_alive, n_step, alive, rays_t, d_ro, d_rd, 1.0, 1 / 256, max_steps, 1, 128, d_grid, nears, fars, xyzs, dirs, deltas,
                      torch.rand(n_alive, device="cuda"))
        sig = (xyzs.abs().sum(1) * 30 + 5).contiguous()
        rgb = (xyzs * 0.5 + 0.5).clamp(0, 1).contiguous()
        z = torch.zeros(n_alive * n_step, device="cuda")
        rm.composite_rays_triplane(n_alive, n_step, 1e-4, alive, rays_t, sig, rgb, deltas, z, z, z, acc["weights_sum"], acc["depth"], image,
                                   acc["amb_aud_sum"], acc["amb_eye_sum"], acc["uncertainty_sum"])
        step += n_step
    w = acc["weights_sum"].cpu().numpy()
    assert w.min() >= 0 and w.max() <= 1 + 1e-5 and (w > 0.5).mean() > 0.3
    img = image.cpu().numpy()
    assert img.min() >= 0 and img.max() <= 1 + 1e-5
    hit = w > 1e-3
    dep = acc["depth"].cpu().numpy()[hit] / w[hit]
    assert (dep >= nears.cpu().numpy()[hit] - 1e-4).all() and (dep <= fars.cpu().numpy()[hit] + 0.05).all()


# ---- radiance field (a18-a20): tri-plane features + attention + sigma / colour MLPs ---------------------------------------------
def _field_case(M, seed, bound=1.0):
    from mere_fusion_amd import weights as W
    from mere_fusion_amd.ernerf.field import grid_geometry
    offsets, pls = grid_geometry(desired_resolution=512 * bound)
    sd = W.make_ernerf_field_state_dict(int(offsets[-1]), seed)
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(M, 3, generator=g) * 2 - 1) * torch.tensor([1.0, 0.5, 1.0]) * bound
    d = torch.randn(M, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    enc_a = torch.randn(1, 32, generator=g)
    c = torch.randn(1, 4, generator=g) * 0.1
    e = torch.tensor([[0.4]])
    return sd, offsets, float(np.log2(pls)), x, d, enc_a, c, e


def test_oracle_field_shapes_and_structure(ref):
    from oracle import ernerf_net_ref as NR
    sd, offsets, S, x, d, enc_a, c, e = _field_case(300, 0)
    assert offsets[-1] == sd["encoder_xy.embeddings"].shape[0] and offsets.shape == (13,)
    sigma, color, aa, ae, unc = NR.field_forward(sd, x, d, enc_a, c, e, offsets, S)
    assert sigma.shape == (300,) and color.shape == (300, 3) and aa.shape == (300, 1) and ae.shape == (300, 1)
    assert (sigma > 0).all() and (color > -0.001 - 1e-6).all() and (color < 1.001 + 1e-6).all()
    assert ((ae > 0) & (ae < 1)).all() and torch.allclose(unc, torch.full((300, 1), float(np.log(2.0))))
    # the audio branch only enters through enc_a * aud_ch_att: a zero audio feature must equal dropping those 32 columns
    s0 = NR.field_forward(sd, x, d, torch.zeros(1, 32), c, e, offsets, S)[0]
    sd2 = dict(sd); sd2["sigma_net.net.0.weight"] = sd["sigma_net.net.0.weight"].clone(); sd2["sigma_net.net.0.weight"][:, 36:68] = 0
    assert torch.allclose(s0, NR.field_forward(sd2, x, d, enc_a, c, e, offsets, S)[0], rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 1000, 1024, 5000, 262144])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_hip_field_matches_oracle(lib_built, ref, M, precision):
    from mere_fusion_amd.ernerf.field import HipNeRFField
    from oracle import ernerf_net_ref as NR
    if M > 10000 and precision == "bf16":
        pytest.skip("full-size case once")
    sd, offsets, S, x, d, enc_a, c, e = _field_case(M, M % 97)
    f = HipNeRFField(sd, bound=1.0, individual_dim=4, exp_eye=True, precision=precision, max_samples=max(M, 2048))
    got = f.forward(x.cuda(), d.cuda(), enc_a.cuda(), c.cuda(), e.cuda())
    n = min(M, 20000)                                                         # oracle on a leading slice of the big case
    want = NR.field_forward(sd, x[:n], d[:n], enc_a, c, e, offsets, S)
    # the reference runs these MLPs in fp16 (autocast, utils.py:1200); bf16x3 is tighter than that, plain bf16 comparable
    tol = 2e-4 if precision == "bf16x3" else 6e-2
    ls_got, ls_want = torch.log(got[0][:n].cpu()), torch.log(want[0])
    assert (ls_got - ls_want).abs().max() <= tol * 4, "log sigma"
    assert (got[1][:n].cpu() - want[1]).abs().max() <= tol, "rgb"
    assert ((got[2][:n].cpu() - want[2]).abs() / (1 + want[2].abs())).max() <= tol, "ambient_aud"
    assert (got[3][:n].cpu() - want[3]).abs().max() <= tol, "ambient_eye"
    assert torch.equal(got[4].cpu(), torch.full((M, 1), float(np.float32(np.log(2.0)))))
    if M >= 5000:
        assert torch.isfinite(got[0]).all() and torch.isfinite(got[1]).all()


# ---- a15: the head render loop ---------------------------------------------------------------------------------------------------
def _sphere_bitfield(H=128, radius=0.45):
    from mere_fusion_amd import weights as W
    return W.make_ernerf_sphere_bitfield(H, radius)


def _camera_rays(W_):
    from mere_fusion_amd import weights as W
    return W.make_ernerf_camera_rays(W_)


def test_sphere_bitfield_matches_oracle_morton(ref):
    bf = _sphere_bitfield(32, 0.5)
    idx = ref.ref_morton3d(16, 16, 16)
    assert bf[idx // 8] & (1 << (idx % 8))                                      # centre voxel occupied
    idx = ref.ref_morton3d(0, 31, 5)
    assert not (bf[idx // 8] & (1 << (idx % 8)))                                # corner voxel empty


@pytest.mark.gpu
@pytest.mark.parametrize("W", [48, 512])
def test_hip_render_loop_matches_oracle(lib_built, ref, W):
    from mere_fusion_amd.ernerf.field import HipNeRFField
    from mere_fusion_amd.ernerf.renderer import HipHeadRenderer
    from oracle import ernerf_render_ref as RR
    sd, offsets, S, _, _, enc_a, c, e = _field_case(8, 3)
    sd = {k: (v * 0.35 if k.startswith("sigma_net.net.2") else v) for k, v in sd.items()}          # keep sigma = exp(h0) in a sane range
    bitfield = _sphere_bitfield()
    ro, rd = _camera_rays(W)
    field = HipNeRFField(sd, max_samples=W * W)
    r = HipHeadRenderer(field, torch.from_numpy(bitfield).cuda(), density_scale=40.0)
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    got = r.run_cuda(_cu(ro), _cu(rd), enc_a.cuda(), c.cuda(), e.cuda(), bg_color=bg, want_u8=True)
    img = got["image"].cpu().numpy()
    w = got["weights_sum"].cpu().numpy()
    assert img.min() >= 0 and img.max() <= 1 and w.max() <= 1 + 1e-5
    hit = w > 0.5
    assert 0.02 < hit.mean() < 0.6                                               # the ball covers part of the view
    np.testing.assert_allclose(img[w == 0], np.broadcast_to([0.1, 0.2, 0.3], img[w == 0].shape), atol=1e-6)   # misses show the background
    assert np.array_equal(got["frame_u8"].cpu().numpy(), (img * 255).astype(np.uint8))           # nerfreal.py:111 truncation
    assert sum(a * s for a, s in got["trace"]) <= 16 * W * W
    if W > 64:
        return                                                                  # the CPU restatement of the full frame takes minutes
    want = RR.run_cuda(sd, offsets, S, ro, rd, enc_a, c, e, bitfield, bg_color=np.array([0.1, 0.2, 0.3], np.float32), density_scale=40.0)
    assert [t[1] for t in got["trace"]] == [t[1] for t in want["trace"]]
    assert all(abs(a[0] - b[0]) <= max(2, 0.002 * b[0]) for a, b in zip(got["trace"], want["trace"]))   # T_thresh ties may move a ray
    # gate on the MAX over all rays.  A ray may stop one sample earlier / later than in the oracle when its transmittance sits within float
    # noise of T_thresh (raymarching.cu:2236): such rays are identified by their weight sums (the skipped sample carries w < T_thresh = 1e-4,
    # float noise is ~1e-6), COUNTED, and still held to the same bound -- the sample they gain or lose is worth < 1e-4 of colour.
    err = np.abs(img - want["image"]).max(1)
    ties = np.abs(w - want["weights_sum"]) > 2e-5
    print(f"render {W}x{W}: image L-inf max {err.max():.3e}, T_thresh tie rays {int(ties.sum())} of {err.size}")
    assert ties.mean() <= 2e-3, int(ties.sum())
    assert err.max() <= 1e-3, (err.max(), int(ties.sum()))
    derr = np.abs(got["depth"].cpu().numpy() - want["depth"])
    assert derr.max() <= 1e-3, derr.max()


# ---- goldens produced by the REFERENCE's own Python (tests/golden/make_ernerf_golden.py) -------------------------------------------
@pytest.fixture(scope="module")
def nerf_golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "ernerf_golden.npz"))


def _golden_field_sd(g):
    from mere_fusion_amd import weights as W
    sd = W.make_ernerf_field_state_dict(int(g["offsets"][-1]), 0)
    return {k: (v * 0.35 if k.startswith("sigma_net.net.2") else v) for k, v in sd.items()}


def test_oracle_field_matches_reference_golden(ref, nerf_golden):
    """Pins oracle/ernerf_net_ref.py to `NeRFNetwork.forward` as the reference executes it (above the extension boundary)."""
    from mere_fusion_amd.ernerf.field import grid_geometry
    from oracle import ernerf_net_ref as NR
    g = nerf_golden
    offsets, pls = grid_geometry()
    assert np.array_equal(offsets, g["offsets"]) and np.float32(np.log2(pls)) == g["log2_per_level_scale"]     # GridEncoder.__init__
    t = lambda k: torch.from_numpy(g[k])
    got = NR.field_forward(_golden_field_sd(g), t("field_x"), t("field_d"), t("field_enc_a"), t("field_c"), t("field_e"), offsets,
                           float(g["log2_per_level_scale"]))
    np.testing.assert_allclose(got[0].numpy(), g["field_sigma"], rtol=2e-5)
    np.testing.assert_allclose(got[1].numpy(), g["field_color"], atol=2e-6)
    np.testing.assert_allclose(got[2].numpy(), g["field_amb_aud"], rtol=2e-5)
    np.testing.assert_allclose(got[3].numpy(), g["field_amb_eye"], atol=2e-6)
    assert list(g["field_unc_shape"]) == [400, 36, 1] and np.allclose(g["field_unc_first"], np.log(2.0))            # the zeros_like(enc_x) quirk


def test_oracle_render_loop_matches_reference_golden(ref, nerf_golden):
    """Pins oracle/ernerf_render_ref.py to the inference branch of the reference's `run_cuda` (renderer.py:231-291)."""
    from mere_fusion_amd import weights as W
    from oracle import ernerf_render_ref as RR
    g = nerf_golden
    Wd = int(g["render_W"])
    ro, rd = W.make_ernerf_camera_rays(Wd)
    want = RR.run_cuda(_golden_field_sd(g), g["offsets"], float(g["log2_per_level_scale"]), ro, rd, torch.from_numpy(g["enc_audio"]),
                       torch.from_numpy(g["render_ind_code"])[None], torch.from_numpy(g["field_e"]), W.make_ernerf_sphere_bitfield(),
                       bg_color=np.array([0.1, 0.2, 0.3], np.float32), density_scale=40.0)
    np.testing.assert_allclose(want["image"], g["render_image"], atol=2e-6)
    np.testing.assert_allclose(want["depth"], g["render_depth"], atol=2e-6)
    np.testing.assert_allclose(want["ambient_aud"], g["render_amb_aud"], rtol=1e-5, atol=1e-5)
    assert (g["render_image"].std(0) > 0.01).all()                              # the golden frame is not flat


@pytest.mark.gpu
def test_hip_field_and_render_match_reference_golden(lib_built, nerf_golden):
    from mere_fusion_amd import weights as W
    from mere_fusion_amd.ernerf.field import HipNeRFField
    from mere_fusion_amd.ernerf.renderer import HipHeadRenderer
    g = nerf_golden
    cu = lambda k: torch.from_numpy(g[k]).cuda()
    field = HipNeRFField(_golden_field_sd(g), max_samples=4096)
    sig, rgb, aa, ae, un = field.forward(cu("field_x"), cu("field_d"), cu("field_enc_a"), cu("field_c"), cu("field_e"))
    assert (torch.log(sig.cpu()) - torch.log(torch.from_numpy(g["field_sigma"]))).abs().max() <= 8e-4
    assert (rgb.cpu() - torch.from_numpy(g["field_color"])).abs().max() <= 2e-4
    assert (ae.cpu() - torch.from_numpy(g["field_amb_eye"])).abs().max() <= 2e-4
    assert ((aa.cpu() - torch.from_numpy(g["field_amb_aud"])).abs() / (1 + torch.from_numpy(g["field_amb_aud"]))).max() <= 2e-4
    Wd = int(g["render_W"])
    ro, rd = W.make_ernerf_camera_rays(Wd)
    r = HipHeadRenderer(field, torch.from_numpy(W.make_ernerf_sphere_bitfield()).cuda(), density_scale=40.0)
    got = r.run_cuda(_cu(ro), _cu(rd), cu("enc_audio"), torch.from_numpy(g["render_ind_code"])[None].cuda(), cu("field_e"),
                     bg_color=torch.tensor([0.1, 0.2, 0.3], device="cuda"))
    err = np.abs(got["image"].cpu().numpy() - g["render_image"]).max(1)
    derr = np.abs(got["depth"].cpu().numpy() - g["render_depth"])
    print(f"render vs the reference's run_cuda golden: image L-inf max {err.max():.3e}, depth max {derr.max():.3e}")
    assert err.max() <= 1e-3, err.max()                      # max over every ray of the reference's frame, no percentile
    assert derr.max() <= 1e-3, derr.max()


# ---- a23: audio feature nets --------------------------------------------------------------------------------------------------
def _audio_sd(g):
    return {k[len("audio_sd/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("audio_sd/")}


def test_oracle_encode_audio_matches_reference_golden(nerf_golden):
    from oracle import ernerf_net_ref as NR
    g = nerf_golden
    got = NR.encode_audio(_audio_sd(g), torch.from_numpy(g["auds"]))
    assert got.shape == (1, 32)
    np.testing.assert_allclose(got.numpy(), g["enc_audio"], rtol=1e-5, atol=1e-6)      # reference: model.encode_audio(auds)
    assert np.abs(g["enc_audio"]).max() > 0.05


@pytest.mark.gpu
def test_hip_encode_audio_matches_reference_golden(lib_built, nerf_golden):
    from mere_fusion_amd.ernerf.audio import HipAudioEncoder
    from oracle import ernerf_net_ref as NR
    g = nerf_golden
    sd = _audio_sd(g)
    enc = HipAudioEncoder(sd, att=2)
    got = enc.encode_audio(torch.from_numpy(g["auds"]).cuda()).cpu().numpy()
    np.testing.assert_allclose(got, g["enc_audio"], rtol=2e-5, atol=2e-6)              # fp32 both sides, different summation order
    one = HipAudioEncoder(sd, att=0)
    w1 = torch.from_numpy(g["auds"][3:4])
    np.testing.assert_allclose(one.encode_audio(w1.cuda()).cpu().numpy(), NR.encode_audio(sd, w1, att=0).numpy(), rtol=2e-5, atol=2e-6)
    with pytest.raises(RuntimeError, match="windows"):
        enc.encode_audio(torch.zeros(3, 44, 16, device="cuda"))
    assert enc.encode_audio(None) is None
    # the lip-smoothing EMA of renderer.py:190-194 inside the same launch: bit-identical to the torch expression it replaces, over a few frames
    a = torch.from_numpy(g["auds"]).cuda()
    prev_t, prev_k = None, None
    for f in range(3):
        af = a * (1.0 + 0.1 * f)
        raw = enc.encode_audio(af)
        prev_t = raw if prev_t is None else 0.35 * prev_t + (1 - 0.35) * raw
        prev_k = enc.encode_audio_smooth(af, prev_k)
        assert torch.equal(prev_k, prev_t), f
    assert enc.encode_audio_smooth(None, None) is None


# ---- a22: torso branch -----------------------------------------------------------------------------------------------------------
def _torso_sd(g):
    from mere_fusion_amd import weights as W
    return W.make_ernerf_torso_state_dict(int(g["torso_offsets"][-1]), 0)


def test_oracle_torso_matches_reference_golden(ref, nerf_golden):
    """Pins oracle run_torso to the reference's own `run_torso` + `forward_torso` (opt.torso model, extensions backed by the C oracle)."""
    from oracle import ernerf_net_ref as NR
    g = nerf_golden
    got = NR.run_torso(_torso_sd(g), g["torso_bg_coords"], g["torso_pose"], g["torso_bg_in"], g["torso_offsets"], float(g["torso_log2_per_level_scale"]))
    np.testing.assert_allclose(got["bg_color"].numpy(), g["torso_bg_color"], atol=3e-6)
    np.testing.assert_allclose(got["torso_alpha"].numpy(), g["torso_alpha"], atol=3e-6)
    m = got["mask"].numpy()
    assert 0.15 < m.mean() < 0.85 and (g["torso_alpha"][~m] == 0).all() and g["torso_alpha"][m].std() > 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_hip_torso_matches_reference_golden(lib_built, ref, nerf_golden, precision):
    from mere_fusion_amd.ernerf.torso import HipTorso
    g = nerf_golden
    t = HipTorso(_torso_sd(g), torso_shrink=0.8, individual_dim=8, precision=precision, max_pixels=4096)
    got = t.run_torso(torch.from_numpy(g["torso_bg_coords"]).cuda(), torch.from_numpy(g["torso_pose"])[None], torch.from_numpy(g["torso_bg_in"]).cuda())
    tol = 3e-4 if precision == "bf16x3" else 5e-2
    assert np.abs(got["torso_alpha"].cpu().numpy() - g["torso_alpha"]).max() <= tol
    assert np.abs(got["bg_color"].cpu().numpy() - g["torso_bg_color"]).max() <= tol
    # a [3] background and the scalar default take the other two branches of the mix
    b3 = t.run_torso(torch.from_numpy(g["torso_bg_coords"]).cuda(), torch.from_numpy(g["torso_pose"])[None], torch.tensor([0.3, 0.5, 0.7], device="cuda"))
    assert torch.allclose(b3["bg_color"], got["bg_color"], atol=1e-6)
    one = t.run_torso(torch.from_numpy(g["torso_bg_coords"]).cuda(), torch.from_numpy(g["torso_pose"])[None], None)
    a = got["torso_alpha"]
    assert torch.allclose(one["bg_color"] - (1 - a), got["bg_color"] - torch.tensor([0.3, 0.5, 0.7], device="cuda") * (1 - a), atol=1e-5)


@pytest.mark.gpu
def test_hip_torso_full_frame_properties(lib_built):
    from mere_fusion_amd import weights as W
    from mere_fusion_amd.ernerf.field import grid_geometry
    from mere_fusion_amd.ernerf.torso import HipTorso
    offs, _ = grid_geometry(num_levels=16, base_resolution=16, log2_hashmap_size=16, desired_resolution=2048)
    t = HipTorso(W.make_ernerf_torso_state_dict(int(offs[-1]), 1), max_pixels=512 * 512)
    u = (torch.arange(512, dtype=torch.float32) + 0.5) / 512 * 2 - 1
    yy, xx = torch.meshgrid(u, u, indexing="ij")
    r = t.run_torso(torch.stack([xx, yy], -1).reshape(-1, 2).cuda(), torch.eye(4)[None], None)
    a = r["torso_alpha"].cpu().numpy()
    assert a.min() >= -0.001 - 1e-6 and a.max() <= 1.001 + 1e-6 and 0.1 < (a > 0).mean() < 0.9
    img = r["bg_color"].cpu().numpy()
    assert np.isfinite(img).all() and img.min() >= -0.01 and img.max() <= 1.01
    assert np.abs(r["deform"].cpu().numpy()).max() < 0.5


@pytest.mark.gpu
def test_hip_torso_fused_kernel_matches_gemm_chain(lib_built, monkeypatch):
    """k_torso_fused (one fp32 kernel, the default) against the twelve-launch bf16x3 GEMM chain it replaces (MF_TORSO=gemm) on a 256 x 256 frame
    with a [N, 3] background: colours, alpha and the deform field agree to the chain's own precision, and the mask (alpha == 0) is the same set."""
    from mere_fusion_amd import weights as W
    from mere_fusion_amd.ernerf.field import grid_geometry
    from mere_fusion_amd.ernerf.torso import HipTorso
    offs, _ = grid_geometry(num_levels=16, base_resolution=16, log2_hashmap_size=16, desired_resolution=2048)
    sd = W.make_ernerf_torso_state_dict(int(offs[-1]), 2)
    n = 256
    u = (torch.arange(n, dtype=torch.float32) + 0.5) / n * 2 - 1
    yy, xx = torch.meshgrid(u, u, indexing="ij")
    coords = torch.stack([xx, yy], -1).reshape(-1, 2).cuda()
    bg = torch.rand(n * n, 3, generator=torch.Generator().manual_seed(3)).cuda()
    pose = torch.eye(4)[None]
    pose[0, :3, 3] = torch.tensor([0.1, -0.2, 0.3])
    fused = HipTorso(sd, max_pixels=n * n).run_torso(coords, pose, bg)
    monkeypatch.setenv("MF_TORSO", "gemm")
    chain = HipTorso(sd, max_pixels=n * n).run_torso(coords, pose, bg)
    for k in ("bg_color", "torso_alpha", "deform"):
        assert (fused[k] - chain[k]).abs().max().item() <= 3e-4, k
    assert torch.equal(fused["torso_alpha"] == 0, chain["torso_alpha"] == 0)
    assert 0.05 < (fused["torso_alpha"] > 0).float().mean().item() < 0.95


@pytest.mark.gpu
@pytest.mark.parametrize("W", [48, 512])
def test_hip_device_controlled_loop_matches_host_loop(lib_built, W):
    """mf_nerf_head_render (round control on the device, no host sync) against the host-driven loop: every ray sees the same samples,
    so the frames must agree to rounding; also through a captured CUDA graph replayed twice."""
    from mere_fusion_amd.ernerf.field import HipNeRFField
    from mere_fusion_amd.ernerf.renderer import HipHeadRenderer
    sd, offsets, S, _, _, enc_a, c, e = _field_case(8, 3)
    sd = {k: (v * 0.35 if k.startswith("sigma_net.net.2") else v) for k, v in sd.items()}
    ro, rd = _camera_rays(W)
    r = HipHeadRenderer(HipNeRFField(sd, max_samples=W * W), torch.from_numpy(_sphere_bitfield()).cuda(), density_scale=40.0)
    bg = torch.tensor([0.1, 0.2, 0.3], device="cuda")
    args = (_cu(ro), _cu(rd), enc_a.cuda(), c.cuda(), e.cuda())
    want = r.run_cuda(*args, bg_color=bg, want_u8=True)
    got = r.run_cuda_device(*args, bg_color=bg, want_u8=True)
    for k, tol in (("image", 1e-6), ("depth", 1e-6), ("weights_sum", 1e-6)):
        assert (got[k] - want[k]).abs().max().item() <= tol, k
    assert (got["frame_u8"].int() - want["frame_u8"].int()).abs().max().item() <= 1
    os.environ["MF_NERF_MARCH"] = "generic"              # the reference-shaped index arithmetic instead of the one-cascade fast path: same bits
    try:
        gen = r.run_cuda_device(*args, bg_color=bg, want_u8=True)
    finally:
        del os.environ["MF_NERF_MARCH"]
    assert torch.equal(gen["image"], got["image"]) and torch.equal(gen["depth"], got["depth"])
    g1 = {k: v.clone() for k, v in r.run_cuda_device(*args, bg_color=bg, want_u8=True, graph=True).items() if v is not None}
    g2 = r.run_cuda_device(*args, bg_color=bg, want_u8=True, graph=True)
    for k in g1:
        assert torch.equal(g1[k], g2[k]) and (g1[k].float() - got[k].float()).abs().max().item() <= (1 if k == "frame_u8" else 1e-6), k


@pytest.mark.gpu
def test_hip_full_frame_device_loop(lib_built):
    """`render(loop="device")`: head with device-side round control, torso on the head's stream -- the same frame as the host-driven `render` (audio nets,
    EMA, torso mix included), three frames in a row."""
    import bench
    a = bench.ErNeRFRunner("bf16x3", 128, torch.device("cuda:0"), seed=3)
    b = bench.ErNeRFRunner("bf16x3", 128, torch.device("cuda:0"), seed=3)
    for i in range(3):
        auds = torch.randn(8, 44, 16, generator=torch.Generator().manual_seed(i)).cuda()
        want = a.r.render(a.ro, a.rd, auds, a.bg_coords, a.pose, a.d_eye, bg_color=1.0, want_u8=True)
        got = b.r.render(b.ro, b.rd, auds, b.bg_coords, b.pose, b.eye, bg_color=1.0, want_u8=True, loop="device")
        torch.cuda.synchronize()
        assert (got["image"] - want["image"]).abs().max().item() <= 1e-6
        assert (got["depth"] - want["depth"]).abs().max().item() <= 1e-6
        assert (got["frame_u8"].int() - want["frame_u8"].int()).abs().max().item() <= 1
        assert want["frame_u8"].float().std().item() > 5


def _head_frame(r, graph=False):
    """one head frame of a bench.ErNeRFRunner through mf_nerf_head_render, with the three per-ray sums and the round count the frame posted"""
    from mere_fusion_amd import _lib
    out = r.r.run_cuda_device(r.ro, r.rd, r.d_enc_a, r.d_ind, r.eye, bg_color=1.0, want_u8=True, graph=graph)
    N = r.ro.shape[0]
    sums = [torch.empty(N, device=r.ro.device) for _ in range(3)]
    _lib.check(_lib.lib().mf_nerf_head_sums(r.r._head, N, *[C.c_void_p(t.data_ptr()) for t in sums], C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    rounds, err = C.c_int(), C.c_int()
    _lib.check(_lib.lib().mf_nerf_head_last_rounds(r.r._head, C.byref(rounds), C.byref(err)))
    assert err.value == 0
    return [out["image"].clone(), out["depth"].clone(), out["weights_sum"].clone(), out["frame_u8"].clone()] + sums, rounds.value


@pytest.mark.gpu
def test_loop_tail_same_bits_wherever_the_launch_chain_hands_over(lib_built, monkeypatch):
    """mf_nerf_head_render enqueues some rounds as (march, field, composite) launches and ONE tail launch for the rest (k_loop_tail).  Wherever the chain hands
    over -- the tail runs the whole loop, the last rounds, or nothing -- the frame is the same bits as the launch-only loop, with 16 (one per chunk), 3 or 1 tail workgroups
    (a lone workgroup runs every chunk of every round itself; three take chunks by ticket and wait for each other's rounds)."""
    import bench
    r = bench.ErNeRFRunner("bf16x3", 128, torch.device("cuda:0"), seed=3)
    host = r.r.run_cuda(r.ro, r.rd, r.d_enc_a, r.d_ind, r.eye, bg_color=1.0, want_u8=True)
    monkeypatch.setenv("MF_NERF_TAIL_AFTER", "off")
    want, rounds = _head_frame(r)
    assert rounds == len(host["trace"]) >= 4, (rounds, host["trace"])
    assert want[0].std().item() > 0.05
    for after, wgs in (("0", None), ("1", None), ("2", None), ("3", None), (str(rounds), None), ("0", "1"), ("1", "3"), ("2", "1")):
        monkeypatch.setenv("MF_NERF_TAIL_AFTER", after)
        if wgs is None:
            monkeypatch.delenv("MF_NERF_TAIL_WGS", raising=False)
        else:
            monkeypatch.setenv("MF_NERF_TAIL_WGS", wgs)
        got, n = _head_frame(r)
        assert n == rounds, (after, wgs, n, rounds)
        for a, b, name in zip(got, want, ("image", "depth", "weights_sum", "frame_u8", "ambient_aud", "ambient_eye", "uncertainty")):
            assert torch.equal(a, b), (after, wgs, name, (a.float() - b.float()).abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("precision,width,max_steps", [("bf16", 20, 64), ("bf16x3", 33, 5), ("bf16x3", 40, 1)])
def test_loop_tail_other_shapes(lib_built, monkeypatch, precision, width, max_steps):
    """The tail with a ray count that is no multiple of its 512-ray chunks (400, 1 089, 1 600 rays), the one-pass bf16 field, a long loop (max_steps 64: one
    ticket set per possible round) and the shortest ones (5 steps, 1 step: the loop ends by `step >= max_steps` with rays still alive)."""
    import bench
    from mere_fusion_amd import _lib
    r = bench.ErNeRFRunner(precision, width, torch.device("cuda:0"), seed=7)

    def frame(after):
        monkeypatch.setenv("MF_NERF_TAIL_AFTER", after)
        out = r.r.run_cuda_device(r.ro, r.rd, r.d_enc_a, r.d_ind, r.eye, bg_color=1.0, want_u8=True, max_steps=max_steps)
        torch.cuda.synchronize()
        rounds, err = C.c_int(), C.c_int()
        _lib.check(_lib.lib().mf_nerf_head_last_rounds(r.r._head, C.byref(rounds), C.byref(err)))
        assert err.value == 0
        return [out[k].clone() for k in ("image", "depth", "weights_sum", "frame_u8")], rounds.value
    want, rounds = frame("off")
    assert 1 <= rounds <= max_steps
    for after in ("0", "1", "2"):
        got, n = frame(after)
        assert n == rounds
        assert all(torch.equal(a, b) for a, b in zip(got, want)), (after, rounds)


@pytest.mark.gpu
def test_launched_rounds_follow_the_frames_before(lib_built, monkeypatch):
    """Without MF_NERF_TAIL_AFTER: every round goes out as launches until a first frame has posted its round count; after that the chain is that count of rounds
    and the tail launch; a captured graph is keyed on the count.  Same bits all along."""
    import bench
    from mere_fusion_amd import _lib
    monkeypatch.delenv("MF_NERF_TAIL_AFTER", raising=False)
    monkeypatch.delenv("MF_NERF_TAIL_WGS", raising=False)
    r = bench.ErNeRFRunner("bf16x3", 96, torch.device("cuda:0"), seed=5)

    def plan():
        k = C.c_int()
        _lib.check(_lib.lib().mf_nerf_head_plan_rounds(r.r._head, 16, C.byref(k)))
        return k.value
    first, rounds = _head_frame(r)
    assert 0 < rounds < 15
    assert plan() == rounds
    second, n2 = _head_frame(r)
    assert n2 == rounds and all(torch.equal(a, b) for a, b in zip(first, second))
    g1, _ = _head_frame(r, graph=True)
    g2, _ = _head_frame(r, graph=True)
    assert all(torch.equal(a, b) for a, b in zip(first, g1)) and all(torch.equal(a, b) for a, b in zip(first, g2))
    assert [k[-1] for k in r.r._graphs] == [rounds]
    # a pinned count wins over the feedback, -1 returns to it
    _lib.check(_lib.lib().mf_nerf_head_set_rounds(r.r._head, 0))
    third, n3 = _head_frame(r)
    _lib.check(_lib.lib().mf_nerf_head_set_rounds(r.r._head, -1))
    assert n3 == rounds and all(torch.equal(a, b) for a, b in zip(first, third))


_RESIZE_CASES = [(24, 24, 24, 24), (24, 32, 45, 70), (64, 48, 17, 31), (1, 5, 4, 9), (50, 50, 450, 450)]


@pytest.mark.parametrize("h,w,H,W", _RESIZE_CASES)
def test_oracle_resize_matches_torch_interpolate(h, w, H, W):
    """a24: the reference resizes with F.interpolate (utils.py:1208-1209); torch on CPU is that very call."""
    import torch.nn.functional as F
    from oracle import ernerf_render_ref as RR
    g = np.random.default_rng(h * 1000 + W)
    img, dep = g.random((h, w, 3), dtype=np.float32), g.random((h, w), dtype=np.float32)
    want = F.interpolate(torch.from_numpy(img)[None].permute(0, 3, 1, 2), size=(H, W), mode="bilinear").permute(0, 2, 3, 1)[0].numpy()
    want_d = F.interpolate(torch.from_numpy(dep)[None, None], size=(H, W), mode="nearest")[0, 0].numpy()
    got, got_d, u8 = RR.resize_frame(img, dep, H, W)
    assert np.abs(got - want).max() <= 2.4e-7          # 2 ulp at 1.0: the blend itself may or may not be contracted
    assert np.array_equal(got_d, want_d)
    assert np.array_equal(u8, (got * 255).astype(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,H,W", _RESIZE_CASES + [(512, 512, 450, 450)])
def test_hip_resize_frame_matches_oracle(lib_built, h, w, H, W):
    from mere_fusion_amd.ernerf.renderer import HipHeadRenderer
    from oracle import ernerf_render_ref as RR
    g = np.random.default_rng(h + W)
    img, dep = g.random((h, w, 3), dtype=np.float32), g.random((h, w), dtype=np.float32)
    r = HipHeadRenderer.__new__(HipHeadRenderer)
    from mere_fusion_amd import _lib
    r._lib, r._head = _lib.lib(), None
    got = r.resize({"image": _cu(img.reshape(-1, 3)), "depth": _cu(dep.reshape(-1))}, h, w, H, W)
    want, want_d, want_u8 = RR.resize_frame(img, dep, H, W)
    assert np.abs(got["image"].cpu().numpy() - want).max() <= 2.4e-7
    assert np.array_equal(got["depth"].cpu().numpy(), want_d)
    assert np.abs(got["frame_u8"].cpu().numpy().astype(int) - want_u8.astype(int)).max() <= 1


@pytest.mark.gpu
def test_hip_device_loop_edge_cases(lib_built):
    """Ragged ray counts (not a multiple of the 256 / 1024-lane blocks), rays that all miss the box (near = far = FLT_MAX -> background only),
    a single round (max_steps = 1), and a second frame on the same handle with fewer rays (stale control block / alive lists)."""
    from mere_fusion_amd.ernerf.field import HipNeRFField
    from mere_fusion_amd.ernerf.renderer import HipHeadRenderer
    sd, offsets, S, _, _, enc_a, c, e = _field_case(8, 3)
    sd = {k: (v * 0.35 if k.startswith("sigma_net.net.2") else v) for k, v in sd.items()}
    r = HipHeadRenderer(HipNeRFField(sd, max_samples=4096), torch.from_numpy(_sphere_bitfield()).cuda(), density_scale=40.0)
    bg = torch.tensor([0.2, 0.4, 0.6], device="cuda")
    ro, rd = _camera_rays(48)
    for n in (1, 255, 257, 1025, 2304, 1000):
        args = (_cu(ro[:n]), _cu(rd[:n]), enc_a.cuda(), c.cuda(), e.cuda())
        want = r.run_cuda(*args, bg_color=bg)
        got = r.run_cuda_device(*args, bg_color=bg)
        assert (got["image"] - want["image"]).abs().max().item() <= 1e-6, n
        assert (got["depth"] - want["depth"]).abs().max().item() <= 1e-6, n
    # every ray points away from the volume: nothing is sampled, the frame is the background
    away = rd.copy(); away[:, 2] = np.abs(away[:, 2]) + 1.0
    far_o = ro.copy(); far_o[:, 2] = 10.0
    got = r.run_cuda_device(_cu(far_o), _cu(away), enc_a.cuda(), c.cuda(), e.cuda(), bg_color=bg)
    assert torch.allclose(got["image"], bg.expand_as(got["image"]), atol=1e-7) and float(got["weights_sum"].abs().max()) == 0.0
    # one round only
    args = (_cu(ro), _cu(rd), enc_a.cuda(), c.cuda(), e.cuda())
    a1, b1 = r.run_cuda(*args, bg_color=bg, max_steps=1), r.run_cuda_device(*args, bg_color=bg, max_steps=1)
    assert (a1["image"] - b1["image"]).abs().max().item() <= 1e-6
