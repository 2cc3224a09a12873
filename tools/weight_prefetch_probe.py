#!/usr/bin/env python3
"""Is a weight-streaming layer of the UNet (batch 8) held back by where its weights come from?  Times the conv launch alone (mf_conv2d_time: back to back, the
weights stay in the Infinity Cache between launches) and COLD (a 1 GB fill between launches evicts L2 and the Infinity Cache: the weights come from HBM, as
in the step, where 3.4 GB of weights pass between two uses of a layer), and cold with the weights touched by a streaming read just before the launch (what a
prefetch branch of the graph would do).     python tools/weight_prefetch_probe.py      (GPU box)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np
import torch
from mere_fusion_amd import _lib
l = _lib.lib(); _lib.init_device(0)
evict = torch.empty(1 << 28, dtype=torch.float32, device="cuda")          # 1 GB


def one(cin, cout, k, hw, B=8, iters=12):
    g = torch.Generator().manual_seed(cin + cout + k + hw)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.zeros(cout)
    d = _lib.MfConv2dDesc(cin=cin, cout=cout, kh=k, kw=k, stride_h=1, stride_w=1, pad_h=k // 2, pad_w=k // 2, transposed=0, output_padding=0, residual=0, act=0,
                          in_h=hw, in_w=hw, upsample=0)
    h = C.c_void_p()
    _lib.check(l.mf_conv2d_create(C.byref(d), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), None, None, None, None, _lib.PRECISIONS["bf16x3"], C.byref(h)))
    x = torch.randn(B, cin, hw, hw, device="cuda")
    y = torch.empty(B, cout, hw, hw, device="cuda")
    for _ in range(2):
        _lib.check(l.mf_conv2d_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), B, None))
    t = C.c_float()
    _lib.check(l.mf_conv2d_time(h, B, 20, C.byref(t), None))
    warm = t.value * 1e3
    # cold: one timed launch behind an eviction pass, repeated
    cold = []
    for _ in range(iters):
        evict.fill_(1.0)
        torch.cuda.synchronize()
        _lib.check(l.mf_conv2d_time(h, B, 1, C.byref(t), None))
        cold.append(t.value * 1e3)
    mb = cin * cout * k * k * 4 / 1e6
    print(f"{cin:5d} -> {cout:5d} k{k} @{hw:2d}^2 B{B}: weights {mb:6.1f} MB; launch alone, back to back {warm:6.1f} us; behind a 1 GB eviction median {np.median(cold):6.1f} us "
          f"(min {min(cold):.1f}) -> cold / warm {np.median(cold) / warm:.2f}", flush=True)
    l.mf_conv2d_destroy(h)


for shape in ((2560, 1280, 3, 8), (1280, 1280, 3, 8), (1280, 1280, 3, 4), (2560, 1280, 3, 4), (1920, 640, 3, 16), (640, 640, 3, 16), (1280, 1280, 1, 8), (960, 320, 3, 32), (320, 320, 1, 32)):
    one(*shape)
