"""The REFERENCE's own ER-NeRF kernels beside the HIP path (SURVEY 8a rows a16, a17, a19, a21 + the frequency encoder of a22).

oracle/_ref/_raymarching_face.so, _shencoder.so, _freqencoder.so are the reference's raymarching.cu / shencoder.cu / freqencoder.cu, built for
gfx950 by oracle/build_ref_ernerf.py from the sources where they lie under /root/reference (PyTorch-ROCm's hipify + hipcc, the real torch
headers; the build container has the sources, the GPU box only the prebuilt modules).  The product's drop-in modules export the same function
names with the same argument order (that IS the drop-in contract, raymarching.h:7-38, shencoder.h:9, freqencoder.h:7), so every test
calls both modules with the same seeded arguments and compares what they wrote.  This pins the rows that round 1 could only hold against
this repo's C restatement (oracle/ernerf_ref.c).  gridencoder.cu does not build on ROCm 7.2 (see build_ref_ernerf.py) and stays with the
restatement + KATs.

Bar: indices / flags / sample positions / step sizes bit-exact (march_rays, near/far, the composite's termination flags and per-ray t); float accumulators
(the composite's blended sums, the encoders) within a few ulp.  The reference is compiled with hipcc's default FMA contraction (as nvcc contracts it on its
own hardware); the product's march kernels are compiled with contraction OFF and carry explicit fmaf() at exactly the places the reference build fuses
(read off its ISA: profiles/r06_march_rays_reference_isa.txt), and so does the plain-C oracle the CPU tests pin them to -- the restatement follows the
reference, not the other way round."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
AABB = np.array([-1, -1, -1, 1, 1, 1], np.float32)


def _load_ref(name):
    path = os.path.join(REFDIR, name + ".so")
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (python oracle/build_ref_ernerf.py in the build container)")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)       # not registered in sys.modules: the product's shim keeps that name
    spec.loader.exec_module(mod)
    return mod


def _mine(name):
    d = os.path.join(ROOT, "mere-fusion_amd", "dropin")
    if d not in sys.path:
        sys.path.insert(0, d)
    return __import__(name)


def _cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _scene(n_rays, seed, H=128, cascades=1, density=0.3):
    rng = np.random.default_rng(seed)
    ro = (rng.standard_normal((n_rays, 3)) * 0.1 + [0, 0, -2.2]).astype(np.float32)
    target = rng.uniform(-0.9, 0.9, (n_rays, 3)).astype(np.float32) * [1, 0.5, 1]
    rd = target - ro
    rd = (rd / np.linalg.norm(rd, axis=1, keepdims=True)).astype(np.float32)
    grid = (rng.random(cascades * H ** 3 // 8) < density).astype(np.uint8) * rng.integers(1, 256, cascades * H ** 3 // 8).astype(np.uint8)
    return ro, rd, grid


def test_reference_modules_are_the_reference_build():
    """CPU tier: when the build container has the reference, the recipe must have produced the modules the GPU tier loads."""
    if not os.path.isdir("/root/reference/ernerf"):
        pytest.skip("no reference checkout here (GPU box): the prebuilt modules are used")
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import build_ref_ernerf as b
    assert b.build(), "oracle/build_ref_ernerf.py failed"
    for name in b.EXTS:
        assert os.path.exists(os.path.join(REFDIR, name + ".so"))
    assert not os.path.exists(os.path.join(REFDIR, "build")), "translated sources must not stay in the tree"


@pytest.mark.gpu
@pytest.mark.parametrize("n_rays,cascades,n_step", [(4096, 1, 1), (4096, 1, 8), (1000, 2, 3), (262144, 1, 2), (65536, 1, 8)])
def test_near_far_and_march_rays_vs_reference_kernels(lib_built, n_rays, cascades, n_step):
    ref, mine = _load_ref("_raymarching_face"), _mine("_raymarching_face")
    ro, rd, grid = _scene(n_rays, n_rays + cascades, cascades=cascades)
    bound = float(2 ** (cascades - 1))
    d_ro, d_rd, d_aabb, d_grid = _cu(ro), _cu(rd), _cu(AABB * bound), _cu(grid)
    out = {}
    for tag, m in (("ref", ref), ("hip", mine)):
        nears, fars = torch.empty(n_rays, device="cuda"), torch.empty(n_rays, device="cuda")
        m.near_far_from_aabb(d_ro, d_rd, d_aabb, n_rays, 0.05, nears, fars)
        out[tag] = [nears, fars]
    torch.cuda.synchronize()
    # a16: slab test = subtract, divide, min / max: nothing to contract -> bit-exact
    np.testing.assert_array_equal(out["hip"][0].cpu().numpy(), out["ref"][0].cpu().numpy())
    np.testing.assert_array_equal(out["hip"][1].cpu().numpy(), out["ref"][1].cpu().numpy())
    nears, fars = out["ref"]
    rng = np.random.default_rng(7)
    alive = rng.permutation(n_rays).astype(np.int32)[: max(1, n_rays * 3 // 4)]
    na = alive.shape[0]
    noises = rng.random(na, np.float32)
    res = {}
    for tag, m in (("ref", ref), ("hip", mine)):
        xyzs, dirs, deltas = (torch.zeros(na * n_step, k, device="cuda") for k in (3, 3, 2))
        rays_t = nears.clone()
        m.march_rays(na, n_step, _cu(alive), rays_t, d_ro, d_rd, bound, 1 / 256, 16, cascades, 128, d_grid, nears, fars, xyzs, dirs, deltas, _cu(noises))
        res[tag] = [t.cpu().numpy() for t in (xyzs, dirs, deltas)]
    x_r, d_r, dl_r = res["ref"]; x_h, d_h, dl_h = res["hip"]
    assert (dl_r[:, 0] > 0).mean() > 0.05                                           # the scene does produce samples
    # a17: index work -- which steps are samples (the occupancy / DDA control flow), the sample positions o + t d, the step sizes and the running t -- is held
    # BIT-EXACT against the reference's own kernel.  The reference build (hipcc here, nvcc on its own hardware) contracts a * b + c into fused multiply-adds at
    # six places of kernel_march_rays (raymarching.cu:870-927: the noise offset, o + t d, x * mip_rbound + 1, level * H3 + morton, the two nested products of
    # the voxel-exit distances); the product kernels AND the plain-C oracle (oracle/ernerf_ref.c) carry explicit fmaf() at exactly those places and nowhere else
    # (round 6, VERDICT r05 item 5: until then the product followed its own restatement, and up to 4 of 49 152 rays stepped into a neighbouring voxel).
    flags_h, flags_r = (dl_h[:, 0] > 0).reshape(na, n_step), (dl_r[:, 0] > 0).reshape(na, n_step)
    tie_rays = (flags_h != flags_r).any(1)
    exact = float((x_h == x_r).mean())
    print(f"[march_rays vs reference kernel] {na} rays x {n_step}: {int(tie_rays.sum())} rays with different sample flags; xyzs bit-equal on {100 * exact:.4f} % of components")
    assert tie_rays.sum() == 0, f"{int(tie_rays.sum())} rays march differently"
    np.testing.assert_array_equal(d_h, d_r)                                          # dirs are copies
    np.testing.assert_array_equal(x_h, x_r)
    np.testing.assert_array_equal(dl_h, dl_r)


@pytest.mark.gpu
@pytest.mark.parametrize("n_alive,n_step", [(5000, 1), (5000, 4), (100000, 8)])
def test_composite_rays_triplane_vs_reference_kernel(lib_built, n_alive, n_step):
    ref, mine = _load_ref("_raymarching_face"), _mine("_raymarching_face")
    rng = np.random.default_rng(n_alive + n_step)
    N = n_alive * 2
    alive = rng.permutation(N).astype(np.int32)[:n_alive]
    rays_t = rng.random(N, np.float32)
    sig = (rng.random((n_alive, n_step), np.float32) * 40).astype(np.float32)
    rgb = rng.random((n_alive, n_step, 3), np.float32)
    dt = (rng.random((n_alive, n_step), np.float32) * 0.03 + 0.001).astype(np.float32)
    dt[rng.random((n_alive, n_step)) < 0.1] = 0                                      # "no sample" markers
    deltas = np.stack([dt, np.cumsum(dt, 1) + 0.2], -1).astype(np.float32)
    aa, ae, unc = (rng.random((n_alive, n_step), np.float32) for _ in range(3))
    acc = dict(weights_sum=(rng.random(N, np.float32) * 0.999).astype(np.float32), depth=rng.random(N, np.float32), image=rng.random((N, 3), np.float32),
               amb_aud_sum=rng.random(N, np.float32), amb_eye_sum=rng.random(N, np.float32), uncertainty_sum=rng.random(N, np.float32))
    res = {}
    for tag, m in (("ref", ref), ("hip", mine)):
        d_alive, d_t = _cu(alive.copy()), _cu(rays_t.copy())
        d_acc = {k: _cu(v.copy()) for k, v in acc.items()}
        m.composite_rays_triplane(n_alive, n_step, 1e-4, d_alive, d_t, _cu(sig), _cu(rgb), _cu(deltas), _cu(aa), _cu(ae), _cu(unc), d_acc["weights_sum"],
                                  d_acc["depth"], d_acc["image"], d_acc["amb_aud_sum"], d_acc["amb_eye_sum"], d_acc["uncertainty_sum"])
        res[tag] = (d_alive.cpu().numpy(), d_t.cpu().numpy(), {k: v.cpu().numpy() for k, v in d_acc.items()})
    # a21: termination flags and the per-ray t are control flow -> exact; the blended sums differ by __expf's / contraction's last bits
    np.testing.assert_array_equal(res["hip"][0], res["ref"][0])
    np.testing.assert_array_equal(res["hip"][1], res["ref"][1])
    for k in acc:
        np.testing.assert_allclose(res["hip"][2][k], res["ref"][2][k], rtol=2e-6, atol=2e-6, err_msg=k)


@pytest.mark.gpu
def test_sh_and_freq_encoders_vs_reference_kernels(lib_built):
    ref_sh, ref_fq = _load_ref("_shencoder"), _load_ref("_freqencoder")
    sh, fq = _mine("_shencoder"), _mine("_freqencoder")
    rng = np.random.default_rng(3)
    d = rng.standard_normal((50000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for degree in (1, 2, 3, 4):
        o_r, o_h = torch.empty(d.shape[0], degree * degree, device="cuda"), torch.empty(d.shape[0], degree * degree, device="cuda")
        ref_sh.sh_encode_forward(_cu(d), o_r, d.shape[0], 3, degree, None)
        sh.sh_encode_forward(_cu(d), o_h, d.shape[0], 3, degree, None)
        # a19: polynomials in x, y, z with |value| <= ~2; the reference contracts a*b + c: a few ulp
        np.testing.assert_allclose(o_h.cpu().numpy(), o_r.cpu().numpy(), rtol=0, atol=1e-6, err_msg=f"degree {degree}")
    for D, deg in ((2, 8), (6, 3), (3, 10)):                                         # forward_torso: freq(2, 8), freq(6, 3) (network.py)
        x = rng.uniform(-1, 1, (4000, D)).astype(np.float32)
        Cc = D + 2 * D * deg
        o_r, o_h = torch.empty(4000, Cc, device="cuda"), torch.empty(4000, Cc, device="cuda")
        ref_fq.freq_encode_forward(_cu(x), 4000, D, deg, Cc, o_r)
        fq.freq_encode_forward(_cu(x), 4000, D, deg, Cc, o_h)
        g, w = o_h.cpu().numpy(), o_r.cpu().numpy()
        np.testing.assert_array_equal(g[:, :D], w[:, :D])                           # identity columns
        # sin of arguments up to 2^deg: both sides use the device's sine; error scales with the argument
        assert np.abs(g - w).max() <= 5e-4 * (2.0 ** deg / 512) + 2e-6


@pytest.mark.gpu
def test_march_composite_loop_vs_reference_kernels(lib_built):
    """renderer.py:246-270 driven once over the reference's kernels and once over the product's, same synthetic sigma / rgb per sample
    position: the frames must agree, and so must the number of rounds and of samples."""
    ref, mine = _load_ref("_raymarching_face"), _mine("_raymarching_face")
    W = 96
    N = W * W
    ro, rd, _ = _scene(N, 11)
    H = 128
    c = (np.arange(H) + 0.5) / H * 2 - 1
    X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
    occ = (X * X + Y * Y + Z * Z) < 0.45 ** 2
    # Morton-ordered bitfield (raymarching.cu:894-895) built with the REFERENCE's own morton3D
    ii = np.stack(np.meshgrid(np.arange(H), np.arange(H), np.arange(H), indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    idx = torch.empty(H ** 3, dtype=torch.int32, device="cuda")
    ref.morton3D(_cu(ii), H ** 3, idx)
    bits = np.zeros(H ** 3, np.uint8)
    bits[idx.cpu().numpy()] = occ.reshape(-1)
    grid = np.packbits(bits.reshape(-1, 8), axis=1, bitorder="little").reshape(-1)
    d_ro, d_rd, d_grid = _cu(ro), _cu(rd), _cu(grid)
    frames, stats = {}, {}
    for tag, m in (("ref", ref), ("hip", mine)):
        nears, fars = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
        m.near_far_from_aabb(d_ro, d_rd, _cu(AABB), N, 0.05, nears, fars)
        acc = [torch.zeros(N, device="cuda") for _ in range(2)] + [torch.zeros(N, 3, device="cuda")] + [torch.zeros(N, device="cuda") for _ in range(3)]
        ws, depth, image, aas, aes, us = acc
        alive = torch.arange(N, dtype=torch.int32, device="cuda")
        rays_t = nears.clone()
        n_alive, step, total = N, 0, 0
        while step < 16 and n_alive > 0:
            n_step = max(min(N // n_alive, 8), 1)
            xyzs, dirs, deltas = (torch.zeros(n_alive * n_step, k, device="cuda") for k in (3, 3, 2))
            m.march_rays(n_alive, n_step, alive, rays_t, d_ro, d_rd, 1.0, 1 / 256, 16, 1, H, d_grid, nears, fars, xyzs, dirs, deltas, torch.zeros(n_alive, device="cuda"))
            total += int((deltas[:, 0] > 0).sum())
            sig = 30.0 * (1.0 + torch.sin(7.0 * xyzs[:, 0]) * torch.cos(5.0 * xyzs[:, 1]))
            rgb = torch.sigmoid(3.0 * xyzs)
            amb = torch.sigmoid(xyzs[:, 2])
            m.composite_rays_triplane(n_alive, n_step, 1e-4, alive, rays_t, sig.contiguous(), rgb.contiguous(), deltas, amb.contiguous(), amb.contiguous(),
                                      amb.contiguous(), ws, depth, image, aas, aes, us)
            alive = alive[alive >= 0]
            n_alive = alive.shape[0]
            step += n_step
        frames[tag] = (image.cpu().numpy(), ws.cpu().numpy(), depth.cpu().numpy())
        stats[tag] = (step, total)
    assert stats["hip"] == stats["ref"], f"rounds / samples differ: {stats}"
    assert stats["ref"][1] > N                                                       # the sphere is hit
    for g, w, name in zip(frames["hip"], frames["ref"], ("image", "weights_sum", "depth")):
        np.testing.assert_allclose(g, w, rtol=0, atol=5e-6, err_msg=name)
