"""Python face of the mf_net_* builder (include/merefusion.h): buffers are ids, every op is appended in execution order."""
import ctypes as C

import numpy as np
import torch

from .. import _lib


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Net:
    def __init__(self, max_batch=1, precision="bf16x3", device="cuda"):
        if not torch.cuda.is_available():
            raise RuntimeError("the avatar-preparation networks need a HIP device; no CPU path exists here")
        self.device = torch.device(device)
        _lib.init_device(self.device.index if self.device.index is not None else torch.cuda.current_device())
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mf_net_create(int(max_batch), _lib.PRECISIONS[precision], C.byref(h)), "net_create")
        self._h, self.max_batch, self.shape, self._keep = h.value, max_batch, {}, []

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().mf_net_destroy(self._h)
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def buffer(self, Cn, H, W, halo=1):
        with torch.cuda.device(self.device):
            i = _lib.lib().mf_net_buffer(self._h, int(Cn), int(H), int(W), int(halo))
        if i < 0:
            _lib.check(i, "net_buffer")
        self.shape[i] = (Cn, H, W)
        return i

    def _f32(self, t):
        if t is None:
            return None
        t = torch.as_tensor(t).detach().to("cpu", torch.float32).contiguous()
        self._keep.append(t)
        return t

    def conv(self, weight, in_buf, out_buf, stride=1, pad=0, act=0, bias=None, bn=None, in_coff=0, out_coff=0, res_buf=-1, res_coff=0, name="conv"):
        """nn.Conv2d(cin, cout, k, stride, pad) [+ BatchNorm2d (gamma, beta, mean, var)] [+ residual] + act (0 none, 1 ReLU, 2 sigmoid)"""
        w = self._f32(weight)
        cout, cin, kh, kw = w.shape
        d = _lib.MfConv2dDesc(cin=cin, cout=cout, kh=kh, kw=kw, stride_h=stride, stride_w=stride, pad_h=pad, pad_w=pad, act=act,
                              residual=1 if res_buf >= 0 else 0)
        b = self._f32(bias)
        g, be, m, v = (self._f32(t) for t in bn) if bn is not None else (None, None, None, None)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mf_net_conv(self._h, C.byref(d), _p(w), _p(b), _p(g), _p(be), _p(m), _p(v), in_buf, in_coff, out_buf, out_coff,
                                              res_buf, res_coff, name.encode()), f"net_conv({name})")

    def maxpool(self, in_buf, out_buf, k, stride, pad=0):
        _lib.check(_lib.lib().mf_net_maxpool(self._h, in_buf, out_buf, k, stride, pad), "net_maxpool")

    def l2norm(self, in_buf, out_buf, weight, eps=1e-10):
        w = self._f32(weight)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mf_net_l2norm(self._h, in_buf, out_buf, _p(w), w.numel(), C.c_float(eps)), "net_l2norm")

    def global_avgpool(self, in_buf, Cn, out_buf, in_coff=0):
        _lib.check(_lib.lib().mf_net_global_avgpool(self._h, in_buf, in_coff, Cn, out_buf), "net_global_avgpool")

    def scale_add(self, x_buf, Cn, out_buf, s_buf=-1, t_buf=-1, v_buf=-1, x_coff=0, t_coff=0, out_coff=0):
        _lib.check(_lib.lib().mf_net_scale_add(self._h, x_buf, x_coff, Cn, s_buf, t_buf, t_coff, v_buf, out_buf, out_coff), "net_scale_add")

    def upsample_nearest(self, in_buf, out_buf):
        _lib.check(_lib.lib().mf_net_upsample_nearest(self._h, in_buf, out_buf), "net_upsample_nearest")

    # ---- execution -------------------------------------------------------------------------------------------------------------
    def set_input(self, buf, x):
        x = x.to(self.device, torch.float32).contiguous()
        B, Cn = x.shape[0], x.shape[1]
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mf_net_set_input(self._h, buf, _p(x), Cn, B, self._stream()), "net_set_input")
        return B

    def run(self, batch):
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mf_net_run(self._h, batch, self._stream()), "net_run")

    def output(self, buf, Cn, batch, coff=0):
        _, H, W = self.shape[buf]
        out = torch.empty((batch, Cn, H, W), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mf_net_get_output(self._h, buf, coff, Cn, _p(out), batch, self._stream()), "net_get_output")
        return out

    def output_bilinear(self, buf, Cn, batch, H, W, coff=0):
        out = torch.empty((batch, Cn, H, W), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mf_net_get_output_bilinear(self._h, buf, coff, Cn, _p(out), H, W, batch, self._stream()), "net_get_output_bilinear")
        return out

    @property
    def gflop_per_item(self):
        return _lib.lib().mf_net_flops_per_item(self._h) / 1e9
