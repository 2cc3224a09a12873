"""GPU parity of the wide 3x3 convolutions (the MuseTalk VAE decoder's shapes) through the C ABI's single-layer entry point
(mf_conv2d_*): the LDS-weights halo kernel's fat tiles (mf_conv_halo2.hip: 16 x 16 pixels x 256 / 128 channels, picked once a layer has
>= 256 workgroups on a >= 64 x 64 map), their implicit-GEMM twin for small batches, and the 4-phase upsample + 3x3, against a float64
torch convolution of the same inputs.  Tolerance: 1e-3 of the output range for bf16x3 (observed ~7e-6), 3e-2 for bf16."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# (cin, cout, H, W, batch, residual `out += x`, upsample); sizes chosen so that both the fat tiles (first rows) and the fallback paths run,
# with maps that are not multiples of the 16 x 16 patch
CASES = [(128, 128, 96, 100, 8, 1, 0), (256, 256, 70, 90, 8, 0, 0), (512, 512, 64, 64, 8, 0, 0), (256, 128, 72, 88, 10, 0, 0),
         (512, 256, 64, 64, 2, 0, 0), (384, 128, 33, 47, 3, 0, 0), (512, 512, 32, 32, 2, 0, 1), (256, 256, 40, 24, 3, 0, 1)]


@pytest.mark.parametrize("cin,cout,H,W,B,res,up", CASES)
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_wide_conv3x3_matches_fp64(lib_built, cin, cout, H, W, B, res, up, precision):
    from mere_fusion_amd import _lib
    if precision == "bf16" and B > 4:
        pytest.skip("bf16 mode: the small cases cover it")
    l = _lib.lib()
    _lib.init_device(0)
    g = torch.Generator().manual_seed(cin * 7 + H)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    d = _lib.MfConv2dDesc(cin=cin, cout=cout, kh=3, kw=3, stride_h=1, stride_w=1, pad_h=1, pad_w=1, transposed=0, output_padding=0,
                          residual=res, act=1, in_h=H, in_w=W, upsample=up)
    h = C.c_void_p()
    _lib.check(l.mf_conv2d_create(C.byref(d), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), None, None, None, None,
                                  _lib.PRECISIONS[precision], C.byref(h)))
    try:
        oh, ow = C.c_int(), C.c_int()
        l.mf_conv2d_out_shape(h, C.byref(oh), C.byref(ow))
        assert (oh.value, ow.value) == ((2 * H, 2 * W) if up else (H, W))
        x = torch.randn(B, cin, H, W, generator=g).cuda()
        y = torch.empty(B, cout, oh.value, ow.value, device="cuda")
        _lib.check(l.mf_conv2d_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), B, None))
        torch.cuda.synchronize()
        xin = torch.nn.functional.interpolate(x.double(), scale_factor=2.0, mode="nearest") if up else x.double()
        ref = torch.nn.functional.conv2d(xin, w.cuda().double(), b.cuda().double(), padding=1)
        if res:
            ref = ref + x.double()
        ref = torch.relu(ref).float()
        err = float((y - ref).abs().max() / ref.abs().max())
        assert err <= (1e-3 if precision == "bf16x3" else 3e-2), err
        # a second launch at a smaller batch takes the other path of a wide plan (twin implicit GEMM) and must agree too
        nb = max(1, B // 4)
        y2 = torch.empty(nb, cout, oh.value, ow.value, device="cuda")
        _lib.check(l.mf_conv2d_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(y2.data_ptr()), nb, None))
        torch.cuda.synchronize()
        err2 = float((y2 - ref[:nb]).abs().max() / ref.abs().max())
        assert err2 <= (1e-3 if precision == "bf16x3" else 3e-2), err2
    finally:
        l.mf_conv2d_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,hw", [(256, 256, 64), (128, 128, 96), (512, 256, 32)])
def test_f16q_experimental_format_vs_fp64(lib_built, cin, cout, hw):
    """MF_PREC_F16Q (experimental, this test seam only): operands as f16 + FP6 (e2m3, MX block scales) residuals, one f16 MFMA + one block-scaled
    16x16x128 MFMA per tap where bf16x3 issues three.  Expected error ~2.5x bf16x3's (numerics study + tools/mx_gemm_probe.hip)."""
    import ctypes as C
    from mere_fusion_amd import _lib
    l = _lib.lib()
    _lib.init_device(0)
    rng = np.random.default_rng(cin + hw)
    w = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * 0.1)
    x = torch.from_numpy(rng.standard_normal((2, cin, hw, hw)).astype(np.float32))
    x = x * torch.sigmoid(x)                                                       # SiLU-shaped, like the VAE's conv inputs
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1).float()
    errs = {}
    for prec in ("bf16x3", "f16q"):
        d = _lib.MfConv2dDesc(cin=cin, cout=cout, kh=3, kw=3, stride_h=1, stride_w=1, pad_h=1, pad_w=1, act=0, in_h=hw, in_w=hw)
        h = C.c_void_p()
        _lib.check(l.mf_conv2d_create(C.byref(d), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), None, None, None, None, _lib.PRECISIONS[prec], C.byref(h)))
        xd, y = x.cuda(), torch.empty(2, cout, hw, hw, device="cuda")
        _lib.check(l.mf_conv2d_forward(h, C.c_void_p(xd.data_ptr()), C.c_void_p(y.data_ptr()), 2, None))
        torch.cuda.synchronize()
        errs[prec] = float((y.cpu() - ref).abs().max())
        l.mf_conv2d_destroy(h)
    print(f"[f16q {cin}->{cout} @{hw}] L-inf vs fp64: bf16x3 {errs['bf16x3']:.2e}, f16 + FP6 {errs['f16q']:.2e} (max |y| {float(ref.abs().max()):.2f})")
    assert errs["f16q"] <= 3e-4 and errs["f16q"] <= 6 * errs["bf16x3"] + 1e-5
    # layers the format has no kernel for are refused, not approximated
    # (since the implicit-GEMM kernel learnt the format -- test below -- that is: cin not a multiple of 64, or <= 32 output channels)
    for ci, co in ((64, 32), (96, 64)):
        d = _lib.MfConv2dDesc(cin=ci, cout=co, kh=1, kw=1, stride_h=1, stride_w=1, pad_h=0, pad_w=0, act=0, in_h=hw, in_w=hw)
        h = C.c_void_p()
        w1 = torch.zeros(co, ci, 1, 1)
        assert l.mf_conv2d_create(C.byref(d), C.c_void_p(w1.data_ptr()), None, None, None, None, None, _lib.PRECISIONS["f16q"], C.byref(h)) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,k,hw,batch", [(320, 320, 1, 32, 8), (640, 640, 3, 16, 8), (1280, 640, 3, 16, 4), (320, 2560, 1, 32, 2)])
def test_f16q_implicit_gemm_vs_fp64(lib_built, cin, cout, k, hw, batch):
    """The implicit-GEMM kernel in the f16 + FP6 format (k_conv_igemm<..., Q>: one f16 MFMA per 32-deep step + one 16x16x128 FP6 MFMA per 64-deep tile) on
    UNet-shaped layers.  Built and measured in round 3 (no faster than bf16x3 there: the loop is bound by operand delivery, not by the matrix pipe -- DESIGN
    section 4), kept behind the conv2d seam; same error budget as the halo tile."""
    import ctypes as C
    from mere_fusion_amd import _lib
    l = _lib.lib()
    _lib.init_device(0)
    rng = np.random.default_rng(cin + cout + k)
    w = torch.from_numpy((rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (cin * k * k))).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * 0.1)
    x = torch.from_numpy(rng.standard_normal((batch, cin, hw, hw)).astype(np.float32))
    x = x * torch.sigmoid(x)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=k // 2).float()
    errs = {}
    for prec in ("bf16x3", "f16q"):
        d = _lib.MfConv2dDesc(cin=cin, cout=cout, kh=k, kw=k, stride_h=1, stride_w=1, pad_h=k // 2, pad_w=k // 2, act=0, in_h=hw, in_w=hw)
        h = C.c_void_p()
        _lib.check(l.mf_conv2d_create(C.byref(d), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), None, None, None, None, _lib.PRECISIONS[prec], C.byref(h)))
        xd, y = x.cuda(), torch.empty(batch, cout, hw, hw, device="cuda")
        _lib.check(l.mf_conv2d_forward(h, C.c_void_p(xd.data_ptr()), C.c_void_p(y.data_ptr()), batch, None))
        torch.cuda.synchronize()
        errs[prec] = float((y.cpu() - ref).abs().max())
        l.mf_conv2d_destroy(h)
    print(f"[f16q igemm {cin}->{cout} k{k} @{hw}] L-inf vs fp64: bf16x3 {errs['bf16x3']:.2e}, f16 + FP6 {errs['f16q']:.2e} (max |y| {float(ref.abs().max()):.2f})")
    assert errs["f16q"] <= 3e-4 and errs["f16q"] <= 6 * errs["bf16x3"] + 1e-5


# A conv as the producer of a GroupNorm (ConvPlan::out_stats): (cin, cout, k, H, W, batch, residual, upsample, groups) chosen so that every way
# the statistics can come out runs at least once -- the 4-wave implicit-GEMM epilogue (wide Linears on 32 x 32 maps), the split-K combine (small
# maps, deep K), the 8-wave / 128 x 128 tiles and the halo kernels with a statistics pass behind them, the split halo tiles (512 channels at
# 32 x 32), the 4-phase upsample conv, channels-per-group 10 / 20 / 40 (the UNet) and 4 / 8 / 16 (the VAE).
STATS_CASES = [(320, 320, 1, 32, 32, 8, 0, 0, 32), (320, 320, 3, 32, 32, 8, 1, 0, 32), (640, 640, 3, 16, 16, 8, 0, 0, 32),
               (1280, 1280, 3, 8, 8, 8, 1, 0, 32), (1280, 640, 1, 16, 16, 3, 0, 0, 32), (512, 512, 3, 32, 32, 8, 0, 0, 32),
               (256, 256, 3, 64, 64, 8, 0, 0, 32), (256, 256, 3, 24, 40, 2, 0, 1, 32), (640, 320, 3, 32, 32, 8, 0, 0, 32),
               (320, 320, 3, 64, 64, 1, 0, 0, 32)]


@pytest.mark.parametrize("cin,cout,k,H,W,B,res,up,groups", STATS_CASES)
def test_conv_leaves_groupnorm_statistics_of_its_output(lib_built, cin, cout, k, H, W, B, res, up, groups):
    from mere_fusion_amd import _lib
    l = _lib.lib()
    _lib.init_device(0)
    g = torch.Generator().manual_seed(cin + 3 * H + k)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.3
    d = _lib.MfConv2dDesc(cin=cin, cout=cout, kh=k, kw=k, stride_h=1, stride_w=1, pad_h=k // 2, pad_w=k // 2, transposed=0, output_padding=0,
                          residual=res, act=0, in_h=H, in_w=W, upsample=up)
    h = C.c_void_p()
    _lib.check(l.mf_conv2d_create(C.byref(d), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), None, None, None, None,
                                  _lib.PRECISIONS["bf16x3"], C.byref(h)))
    try:
        oh, ow = C.c_int(), C.c_int()
        l.mf_conv2d_out_shape(h, C.byref(oh), C.byref(ow))
        x = torch.randn(B, cin, H, W, generator=g).cuda() + 0.25
        for nb in (B, max(1, B // 2)):                     # the smaller batch usually lands on another tile / split
            y = torch.empty(nb, cout, oh.value, ow.value, device="cuda")
            st = torch.full((nb, groups, 2), 7.0, dtype=torch.float64, device="cuda")
            _lib.check(l.mf_conv2d_forward_stats(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), groups, C.c_void_p(st.data_ptr()), nb, None))
            torch.cuda.synchronize()
            yg = y.double().reshape(nb, groups, -1)
            want = torch.stack([yg.sum(-1), (yg * yg).sum(-1)], dim=-1)
            # the kernels sum the fp32 values they store as (hi, lo) pairs; y is those values back in fp32: agreement to fp32 summation noise
            scale = want.abs().amax(dim=(0, 1)) + 1e-30
            err = float(((st - want).abs() / scale).max())
            assert err <= 2e-6, (nb, err)
            # and the output itself is what the plain launch writes
            y2 = torch.empty_like(y)
            _lib.check(l.mf_conv2d_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(y2.data_ptr()), nb, None))
            torch.cuda.synchronize()
            assert torch.equal(y, y2)
    finally:
        l.mf_conv2d_destroy(h)


@pytest.mark.parametrize("cin,cout,H,W,B", [(256, 256, 64, 64, 2), (512, 512, 32, 48, 3), (128, 256, 40, 24, 2)])
def test_f16q_upsample_phases_vs_fp64(lib_built, cin, cout, H, W, B):
    """nearest 2x upsample + 3x3 in the f16 + FP6 format (the VAE decoder's upsamplers): four 2 x 2-tap phase launches of the halo tile, pre-summed taps
    quantised per phase; against a float64 convolution of the upsampled input, beside the bf16x3 4-phase implicit GEMM; statistics seam included."""
    from mere_fusion_amd import _lib
    l = _lib.lib()
    _lib.init_device(0)
    rng = np.random.default_rng(cin + H)
    w = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32) * 0.1)
    x = torch.from_numpy(rng.standard_normal((B, cin, H, W)).astype(np.float32))
    ref = torch.nn.functional.conv2d(torch.nn.functional.interpolate(x.double(), scale_factor=2.0, mode="nearest"), w.double(), b.double(), padding=1).float()
    errs = {}
    for prec in ("bf16x3", "f16q"):
        d = _lib.MfConv2dDesc(cin=cin, cout=cout, kh=3, kw=3, stride_h=1, stride_w=1, pad_h=1, pad_w=1, act=0, in_h=H, in_w=W, upsample=1)
        h = C.c_void_p()
        _lib.check(l.mf_conv2d_create(C.byref(d), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), None, None, None, None, _lib.PRECISIONS[prec], C.byref(h)))
        xd, y = x.cuda(), torch.empty(B, cout, 2 * H, 2 * W, device="cuda")
        st = torch.zeros(B, 32, 2, dtype=torch.float64, device="cuda")
        _lib.check(l.mf_conv2d_forward_stats(h, C.c_void_p(xd.data_ptr()), C.c_void_p(y.data_ptr()), 32, C.c_void_p(st.data_ptr()), B, None))
        torch.cuda.synchronize()
        errs[prec] = float((y.cpu() - ref).abs().max())
        yg = y.double().reshape(B, 32, -1)
        want = torch.stack([yg.sum(-1), (yg * yg).sum(-1)], dim=-1)
        serr = float(((st - want).abs() / (want.abs().amax(dim=(0, 1)) + 1e-30)).max())
        assert serr <= 2e-6, (prec, serr)
        l.mf_conv2d_destroy(h)
    print(f"[f16q upsample {cin}->{cout} @{H}x{W}] L-inf vs fp64: bf16x3 {errs['bf16x3']:.2e}, f16 + FP6 {errs['f16q']:.2e} (max |y| {float(ref.abs().max()):.2f})")
    assert errs["f16q"] <= 3e-4 and errs["f16q"] <= 6 * errs["bf16x3"] + 1e-5

# Channel counts that are multiples of 80 (the UNet's 320 / 640 / 1280): the shapes the 128 x 80 producer-wave tile can take (256 workgroups for 320 channels over
# 8192 pixels).  Run as is this checks whatever the tuning table picks; tests/test_conv_producer_waves.py re-runs it with that tile forced.  Ragged pixel counts,
# contractions that are not multiples of 64, one and several channel tiles, residual on / off.
C80_CASES = [(320, 320, 3, 32, 32, 8, 1), (72, 160, 3, 13, 13, 8, 0), (640, 320, 1, 32, 32, 8, 0), (100, 80, 3, 9, 11, 5, 0), (1280, 640, 3, 16, 16, 3, 0),
             (320, 320, 1, 32, 32, 2, 1)]


@pytest.mark.parametrize("cin,cout,k,H,W,B,res", C80_CASES)
def test_channel_multiples_of_80_vs_fp64(lib_built, cin, cout, k, H, W, B, res):
    from mere_fusion_amd import _lib
    l = _lib.lib()
    _lib.init_device(0)
    g = torch.Generator().manual_seed(cin + 5 * H + k)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.2
    d = _lib.MfConv2dDesc(cin=cin, cout=cout, kh=k, kw=k, stride_h=1, stride_w=1, pad_h=k // 2, pad_w=k // 2, transposed=0, output_padding=0,
                          residual=res, act=1, in_h=H, in_w=W, upsample=0)
    h = C.c_void_p()
    _lib.check(l.mf_conv2d_create(C.byref(d), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), None, None, None, None,
                                  _lib.PRECISIONS["bf16x3"], C.byref(h)))
    try:
        x = torch.randn(B, cin, H, W, generator=g).cuda()
        ref = torch.nn.functional.conv2d(x.double(), w.cuda().double(), b.cuda().double(), padding=k // 2)
        if res:
            ref = ref + x.double()
        ref = torch.relu(ref).float()
        for nb in (B, 1):
            y = torch.empty(nb, cout, H, W, device="cuda")
            _lib.check(l.mf_conv2d_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), nb, None))
            torch.cuda.synchronize()
            err = float((y - ref[:nb]).abs().max() / ref.abs().max())
            assert err <= 1e-4, (nb, err)                # bf16x3: observed ~5e-6
    finally:
        l.mf_conv2d_destroy(h)


THIN_CASES = [(6, 16, 7, 1, 50, 37, 3), (16, 32, 3, 2, 96, 96, 4), (12, 32, 3, 2, 33, 47, 2), (1, 32, 3, 1, 80, 16, 5), (3, 20, 3, 1, 40, 40, 2), (8, 16, 3, 2, 32, 32, 1)]


@pytest.mark.parametrize("cin,cout,k,stride,H,W,B", THIN_CASES)
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_thin_input_conv_matches_fp64(lib_built, monkeypatch, cin, cout, k, stride, H, W, B, precision):
    """Convolutions with <= 16 input channels (Wav2Lip's first face-encoder layers, wav2lip.py:19-21; the audio encoder's first layer) on k_conv_thin -- the input
    patch in LDS once, every tap read out of it -- and on the implicit GEMM they used to take (MF_CONV_THIN=0), both against a float64 convolution: k7 s1, k3 s1,
    k3 s2, odd map sizes, channel counts that are no multiple of 8 / 16, one and two output fragments."""
    from mere_fusion_amd import _lib
    l = _lib.lib()
    _lib.init_device(0)
    g = torch.Generator().manual_seed(cin * 11 + H)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(B, cin, H, W, generator=g).cuda()
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.cuda().double(), b.cuda().double(), stride=stride, padding=k // 2)).float()
    for mode in ("1", "0"):
        monkeypatch.setenv("MF_CONV_THIN", mode)
        d = _lib.MfConv2dDesc(cin=cin, cout=cout, kh=k, kw=k, stride_h=stride, stride_w=stride, pad_h=k // 2, pad_w=k // 2, transposed=0, output_padding=0, residual=0,
                              act=1, in_h=H, in_w=W)
        h = C.c_void_p()
        _lib.check(l.mf_conv2d_create(C.byref(d), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), None, None, None, None, _lib.PRECISIONS[precision], C.byref(h)))
        try:
            oh, ow = C.c_int(), C.c_int()
            l.mf_conv2d_out_shape(h, C.byref(oh), C.byref(ow))
            assert (oh.value, ow.value) == tuple(ref.shape[2:])
            y = torch.full(tuple(ref.shape), float("nan"), device="cuda")
            _lib.check(l.mf_conv2d_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), B, None))
            torch.cuda.synchronize()
            err = float((y - ref).abs().max() / ref.abs().max())
            assert err <= (1e-3 if precision == "bf16x3" else 3e-2), (mode, err)
        finally:
            l.mf_conv2d_destroy(h)
