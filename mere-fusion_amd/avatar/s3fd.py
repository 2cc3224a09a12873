"""`s3fd` drop-in (face_detection/detection/sfd/net_s3fd.py:22-129; identical copies under wav2lip/ and musetalk/utils/).

`SFDDetector.__init__` does `self.face_detector = s3fd(); .load_state_dict(weights); .to(device); .eval()` and `detect` / `batch_detect`
call `net(img)` on a float tensor [B, 3, H, W] (BGR minus the channel means, sfd/detect.py:19-33,54-66), reading back the list
[cls1, reg1, ..., cls6, reg6].  Same surface here; the 12 tensors stay on the device (the reference's softmax / threshold / decode /
nms code runs on them unchanged).  One graph per input size; conf and loc heads of a level share one convolution."""
import ctypes as C

import torch

from .. import _lib
from .net import Net

_TRUNK = [("conv1_1", 3, 64), ("conv1_2", 64, 64), "pool", ("conv2_1", 64, 128), ("conv2_2", 128, 128), "pool",
          ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), "pool",
          ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512), "pool",
          ("conv5_1", 512, 512), ("conv5_2", 512, 512), ("conv5_3", 512, 512), "pool"]


class s3fd:
    def __init__(self, precision="bf16x3", max_batch=16, device="cuda"):
        self.precision, self.max_batch, self.device = precision, max_batch, torch.device(device)
        self._sd, self._nets = None, {}

    def load_state_dict(self, sd, strict=True):
        self._sd = {k: v.detach().to("cpu", torch.float32) for k, v in sd.items()}
        self._nets = {}
        return self

    def to(self, device):
        return self

    def eval(self):
        return self

    def _build(self, H, W):
        sd = self._sd
        n = Net(self.max_batch, self.precision, self.device)
        g = dict(net=n, taps={})
        x = g["inp"] = n.buffer(3, H, W, 1)
        h, w = H, W
        for idx, item in enumerate(_TRUNK):
            if item == "pool":                                           # F.max_pool2d(h, 2, 2)
                h, w = h // 2, w // 2
                if h < 1 or w < 1:
                    raise ValueError(f"s3fd: a {H}x{W} image is too small for the five pooling stages")
                y = n.buffer(n.shape[x][0], h, w, 3 if idx == len(_TRUNK) - 1 else 1)        # fc6 pads by 3
                n.maxpool(x, y, 2, 2)
            else:
                name, ci, co = item
                y = n.buffer(co, h, w, 1)
                n.conv(sd[name + ".weight"], x, y, 1, 1, act=1, bias=sd[name + ".bias"], name=name)
                if name in ("conv3_3", "conv4_3", "conv5_3"):
                    g["taps"][name] = (y, co, h, w)
            x = y
        # fc6: Conv2d(512, 1024, 3, 1, 3) -- padding 3, so the map grows by 4 (the last pool's buffer carries a halo of 3)
        h, w = h + 4, w + 4
        y = n.buffer(1024, h, w, 1); n.conv(sd["fc6.weight"], x, y, 1, 3, act=1, bias=sd["fc6.bias"], name="fc6"); x = y
        y = n.buffer(1024, h, w, 1); n.conv(sd["fc7.weight"], x, y, 1, 0, act=1, bias=sd["fc7.bias"], name="fc7"); x = y
        feats = []
        for name, scale_key in (("conv3_3", "conv3_3_norm"), ("conv4_3", "conv4_3_norm"), ("conv5_3", "conv5_3_norm")):
            b, co, fh, fw = g["taps"][name]
            nb = n.buffer(co, fh, fw, 1)
            n.l2norm(b, nb, sd[scale_key + ".weight"], 1e-10)
            feats.append((scale_key, nb, fh, fw))
        feats.append(("fc7", x, h, w))
        y = n.buffer(256, h, w, 1); n.conv(sd["conv6_1.weight"], x, y, 1, 0, act=1, bias=sd["conv6_1.bias"], name="conv6_1"); x = y
        h, w = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        y = n.buffer(512, h, w, 1); n.conv(sd["conv6_2.weight"], x, y, 2, 1, act=1, bias=sd["conv6_2.bias"], name="conv6_2"); x = y
        feats.append(("conv6_2", x, h, w))
        y = n.buffer(128, h, w, 1); n.conv(sd["conv7_1.weight"], x, y, 1, 0, act=1, bias=sd["conv7_1.bias"], name="conv7_1"); x = y
        h, w = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        y = n.buffer(256, h, w, 1); n.conv(sd["conv7_2.weight"], x, y, 2, 1, act=1, bias=sd["conv7_2.bias"], name="conv7_2"); x = y
        feats.append(("conv7_2", x, h, w))
        # heads: conf | loc of one level as one 3x3 convolution with 8 output channels (conf padded to 4)
        g["heads"] = []
        for key, fb, fh, fw in feats:
            wc, bc = sd[key + "_mbox_conf.weight"], sd[key + "_mbox_conf.bias"]
            wl, bl = sd[key + "_mbox_loc.weight"], sd[key + "_mbox_loc.bias"]
            nc = wc.shape[0]
            wcat = torch.zeros((8,) + tuple(wc.shape[1:]))
            bcat = torch.zeros(8)
            wcat[:nc], wcat[4:8], bcat[:nc], bcat[4:8] = wc, wl, bc, bl
            ob = n.buffer(8, fh, fw, 0)
            n.conv(wcat, fb, ob, 1, 1, act=0, bias=bcat, name=key + "_mbox")
            g["heads"].append((ob, nc))
        return g

    def __call__(self, x):
        if self._sd is None:
            raise RuntimeError("s3fd: load_state_dict first (sfd_detector.py:27-28)")
        x = torch.as_tensor(x)
        B, Cn, H, W = x.shape
        if B > self.max_batch:
            raise ValueError(f"s3fd: batch {B} exceeds max_batch {self.max_batch}")
        g = self._nets.get((H, W))
        if g is None:
            g = self._nets[(H, W)] = self._build(H, W)
        n = g["net"]
        n.set_input(g["inp"], x)
        n.run(B)
        outs = []
        for i, (ob, nc) in enumerate(g["heads"]):
            cls = n.output(ob, nc, B, coff=0)
            if i == 0:                                                   # max-out background label (net_s3fd.py:123-126)
                c2 = torch.empty((B, 2) + tuple(cls.shape[2:]), dtype=torch.float32, device=cls.device)
                _lib.check(_lib.lib().mf_s3fd_maxout_bg(C.c_void_p(cls.data_ptr()), C.c_void_p(c2.data_ptr()), B, cls.shape[2] * cls.shape[3],
                                                        C.c_void_p(torch.cuda.current_stream(cls.device).cuda_stream)), "s3fd_maxout_bg")
                cls = c2
            outs += [cls, n.output(ob, 4, B, coff=4)]
        return outs

    forward = __call__
