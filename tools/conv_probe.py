#!/usr/bin/env python3
"""Times ONE fused conv layer through the C ABI (mf_conv2d_*), for kernel tuning and PMC runs.

    python tools/conv_probe.py --cin 64 --cout 64 --hw 96 --batch 16 --precision bf16x3 [--iters 50]
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np
import torch
from mere_fusion_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=64)
ap.add_argument("--cout", type=int, default=64)
ap.add_argument("--k", type=int, default=3)
ap.add_argument("--stride", type=int, default=1)
ap.add_argument("--pad", type=int, default=1)
ap.add_argument("--hw", type=int, default=96)
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--residual", type=int, default=1)
ap.add_argument("--transposed", type=int, default=0)
ap.add_argument("--outpad", type=int, default=0)
ap.add_argument("--precision", default="bf16x3")
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--alone-iters", type=int, default=0, help="iterations of the kernel-only timing (default: --iters)")
ap.add_argument("--upsample", type=int, default=0, help="1: nearest 2x upsample in front of the conv")
ap.add_argument("--check", type=int, default=0, help="1: compare with torch fp32 conv2d (+ReLU, residual) on the GPU")
a = ap.parse_args()

l = _lib.lib()
_lib.init_device(0)
rng = np.random.default_rng(0)
shape = (a.cin, a.cout, a.k, a.k) if a.transposed else (a.cout, a.cin, a.k, a.k)
w = torch.from_numpy((rng.standard_normal(shape) * np.sqrt(2.0 / (a.cin * a.k * a.k))).astype(np.float32))
b = torch.zeros(a.cout)
d = _lib.MfConv2dDesc(cin=a.cin, cout=a.cout, kh=a.k, kw=a.k, stride_h=a.stride, stride_w=a.stride, pad_h=a.pad,
                      pad_w=a.pad, transposed=a.transposed, output_padding=a.outpad, residual=a.residual, act=1,
                      in_h=a.hw, in_w=a.hw, upsample=a.upsample)
h = C.c_void_p()
_lib.check(l.mf_conv2d_create(C.byref(d), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), None, None, None, None,
                              _lib.PRECISIONS[a.precision], C.byref(h)))
oh, ow = C.c_int(), C.c_int()
l.mf_conv2d_out_shape(h, C.byref(oh), C.byref(ow))
x = torch.randn(a.batch, a.cin, a.hw, a.hw, device="cuda")
y = torch.empty(a.batch, a.cout, oh.value, ow.value, device="cuda")
for _ in range(3):
    _lib.check(l.mf_conv2d_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), a.batch, None))
torch.cuda.synchronize()
if a.check and not a.transposed:
    xin = torch.nn.functional.interpolate(x.double(), scale_factor=2.0, mode="nearest") if a.upsample else x.double()
    ref = torch.nn.functional.conv2d(xin, w.cuda().double(), b.cuda().double(), stride=a.stride, padding=a.pad)
    if a.residual:
        ref = ref + x.double()
    ref = torch.relu(ref).float()
    print(f"   check vs torch fp64 conv: L-inf {float((y - ref).abs().max()):.3e}, rel {float((y - ref).abs().max() / ref.abs().max()):.3e}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    _lib.check(l.mf_conv2d_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), a.batch, None))
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
taps = a.k * a.k
sites = a.hw * a.hw if a.transposed else oh.value * ow.value   # (upsample: counted at the output resolution, 9 taps)
gf = 2.0 * a.batch * sites * a.cin * a.cout * taps / 1e9
t = C.c_float()
_lib.check(l.mf_conv2d_time(h, a.batch, a.alone_iters or a.iters, C.byref(t), None))
print(f"   conv launch alone: {t.value * 1e3:.1f} us -> {gf / t.value:.1f} TFLOP/s algorithmic")
print(f"conv {a.cin}->{a.cout} k{a.k} s{a.stride} @{a.hw}^2 B{a.batch} {a.precision}: {ms * 1e3:.1f} us per forward "
      f"(incl. nchw<->nhwc passes), {gf:.2f} GF")
