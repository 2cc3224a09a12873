#!/usr/bin/env python3
"""One conv layer (mf_conv2d_*) on a batch of IDENTICAL images: are the outputs of the copies bit-identical?  (They must be: a row's result may not depend on where
its image sits in the batch.)   python tools/conv_copies_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import torch
from mere_fusion_amd import _lib
l = _lib.lib(); _lib.init_device(0)
def run(cin, cout, k, hw, B, res, act, stats_groups=0):
    g = torch.Generator().manual_seed(cin + cout + k)
    w = torch.randn(cout, cin, k, k, generator=g) * (2.0 / (cin * k * k)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    d = _lib.MfConv2dDesc(cin=cin, cout=cout, kh=k, kw=k, stride_h=1, stride_w=1, pad_h=k // 2, pad_w=k // 2, transposed=0, output_padding=0,
                          residual=res, act=act, in_h=hw, in_w=hw, upsample=0)
    h = C.c_void_p()
    _lib.check(l.mf_conv2d_create(C.byref(d), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), None, None, None, None, _lib.PRECISIONS["bf16x3"], C.byref(h)))
    x = torch.randn(1, cin, hw, hw, generator=g).repeat(B, 1, 1, 1).cuda().contiguous()
    co = cout // 2 if act == 5 else cout
    y = torch.empty(B, co, hw, hw, device="cuda")
    if stats_groups:
        st = torch.zeros(B, stats_groups, 2, dtype=torch.float64, device="cuda")
        _lib.check(l.mf_conv2d_forward_stats(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), stats_groups, C.c_void_p(st.data_ptr()), B, None))
    else:
        _lib.check(l.mf_conv2d_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), B, None))
    torch.cuda.synchronize()
    diff = [(y[i] - y[0]).abs().max().item() for i in range(B)]
    sd = [(st[i] - st[0]).abs().max().item() for i in range(B)] if stats_groups else None
    print(f"{cin}->{cout} k{k} @{hw}^2 B{B} res{res} act{act} stats{stats_groups}: copies vs copy 0:", ["%.1e" % v for v in diff], ("stats " + str(["%.1e" % v for v in sd])) if sd else "")
    l.mf_conv2d_destroy(h)
for B in (8, 4):
    run(1280, 1280, 3, 4, B, 1, 0)
    run(1280, 1280, 3, 4, B, 0, 0, 32)
    run(1280, 1280, 1, 4, B, 0, 0)
    run(1280, 10240, 1, 4, B, 0, 5)
    run(2560, 1280, 3, 4, B, 0, 0, 32)
    run(1280, 1280, 3, 8, B, 1, 0, 32)
    run(640, 640, 3, 16, B, 1, 0, 32)
