#!/usr/bin/env python3
"""Do small conv launches on two HIP streams overlap on this box?  Times N launches of one small fused
conv layer on one stream vs the same N split over two streams (two handles)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np
import torch
from mere_fusion_amd import _lib

l = _lib.lib()
_lib.init_device(0)
hw, cin, cout, B, N = int(os.environ.get("HW", 24)), 64, 64, 16, 400


def make():
    w = torch.randn(cout, cin, 3, 3) * 0.05
    b = torch.zeros(cout)
    d = _lib.MfConv2dDesc(cin=cin, cout=cout, kh=3, kw=3, stride_h=1, stride_w=1, pad_h=1, pad_w=1, transposed=0,
                          output_padding=0, residual=1, act=1, in_h=hw, in_w=hw)
    h = C.c_void_p()
    _lib.check(l.mf_conv2d_create(C.byref(d), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), None, None, None, None, 1, C.byref(h)))
    x = torch.randn(B, cin, hw, hw, device="cuda")
    y = torch.empty(B, cout, hw, hw, device="cuda")
    _lib.check(l.mf_conv2d_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), B, None))
    torch.cuda.synchronize()
    return h


h1, h2 = make(), make()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
t = C.c_float()


def run(streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # mf_conv2d_time enqueues `iters` conv launches back to back on the stream; it blocks on its own end
    # event, so drive the two streams from two host threads
    import threading
    th = [threading.Thread(target=lambda hh=hh, ss=ss, n=n: l.mf_conv2d_time(hh, B, n, C.byref(C.c_float()), C.c_void_p(ss.cuda_stream)))
          for hh, ss, n in streams]
    for x in th:
        x.start()
    for x in th:
        x.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6


for _ in range(2):
    one = run([(h1, s1, N)])
    two = run([(h1, s1, N // 2), (h2, s2, N // 2)])
    print(f"{hw}x{hw}: {N} launches on one stream {one:.0f} us ({one / N:.1f} us each); split over two streams {two:.0f} us "
          f"-> overlap factor {one / two:.2f}")
