"""GPU box: what the GEGLU epilogue costs.  ff.net.0.proj of the UNet's transformer blocks (Linear C -> 8 C with `value * gelu(gate)`, 16 launches of ~62 us per step at
batch 8) as a 1 x 1 convolution through mf_conv2d_*: the same GEMM with a plain epilogue (act 0: all 8 C columns stored) and with the GEGLU epilogue (act 5: 4 C stored)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mere_fusion_amd import _lib
l = _lib.lib(); _lib.init_device(0)
print("| layer (tokens x K -> N) | act 0 us | GEGLU us |\n|---|---:|---:|")
for Cc, hw in ((320, 32), (640, 16), (1280, 8)):
    g = torch.Generator().manual_seed(Cc)
    w = torch.randn(8 * Cc, Cc, 1, 1, generator=g) * (1.0 / Cc) ** 0.5
    b = torch.randn(8 * Cc, generator=g) * 0.1
    x = torch.randn(8, Cc, hw, hw, generator=g).cuda()
    res = {}
    for act in (0, 5):
        d = _lib.MfConv2dDesc(cin=Cc, cout=8 * Cc, kh=1, kw=1, stride_h=1, stride_w=1, pad_h=0, pad_w=0, transposed=0, output_padding=0, residual=0, act=act, in_h=hw, in_w=hw)
        h = C.c_void_p()
        _lib.check(l.mf_conv2d_create(C.byref(d), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), None, None, None, None, _lib.PRECISIONS["bf16x3"], C.byref(h)))
        y = torch.empty(8, (4 if act == 5 else 8) * Cc, hw, hw, device="cuda")
        _lib.check(l.mf_conv2d_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), 8, None))
        ms = C.c_float()
        _lib.check(l.mf_conv2d_time(h, 8, 50, C.byref(ms), None))
        res[act] = ms.value * 1e3
        l.mf_conv2d_destroy(h)
    print(f"| {8 * hw * hw} x {Cc} -> {8 * Cc} | {res[0]:.1f} | {res[5]:.1f} |", flush=True)
