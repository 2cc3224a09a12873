"""`BiSeNet` drop-in (musetalk/utils/face_parsing/model.py:236-262 over resnet.py:60-95).

`FaceParsing.model_init` does `net = BiSeNet(resnet_path); net.cuda(); net.load_state_dict(torch.load(model_pth)); net.eval()` and
`__call__` reads `self.net(img)[0]` for a normalised [1, 3, 512, 512] tensor (face_parsing/__init__.py:17-27,51).  Same surface; the
state-dict names are the module tree's (`cp.resnet.layer1.0.conv1.weight`, `cp.arm16.conv_atten.weight`, `ffm.convblk.bn.running_mean`, ...).
`forward` returns (feat_out, feat_out16, feat_out32) like the reference when `aux=True`, else (feat_out,) -- FaceParsing only reads [0]."""
import torch

from .net import Net


class BiSeNet:
    def __init__(self, resnet_path=None, n_classes=19, precision="bf16x3", max_batch=1, device="cuda", aux=False):
        self.n_classes, self.precision, self.max_batch, self.device, self.aux = n_classes, precision, max_batch, torch.device(device), aux
        self._sd, self._nets = None, {}

    def load_state_dict(self, sd, strict=True):
        self._sd = {k: v.detach().to("cpu", torch.float32) for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point()}
        self._nets = {}
        return self

    def cuda(self):
        return self

    def to(self, device):
        return self

    def eval(self):
        return self

    # ---- module tree -> ops ---------------------------------------------------------------------------------------------------
    def _bn(self, p):
        sd = self._sd
        return (sd[p + ".weight"], sd[p + ".bias"], sd[p + ".running_mean"], sd[p + ".running_var"])

    def _cbr(self, n, p, x, y, k, stride, pad, **kw):
        """ConvBNReLU (model.py:12-31)"""
        n.conv(self._sd[p + ".conv.weight"], x, y, stride, pad, act=1, bn=self._bn(p + ".bn"), name=p, **kw)

    def _block(self, n, p, x, cin, cout, stride, h, w, out=None, out_coff=0):
        """BasicBlock (resnet.py:17-47): relu(bn2(conv2(relu(bn1(conv1(x))))) + shortcut)"""
        sd = self._sd
        ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
        t = n.buffer(cout, ho, wo, 1)
        n.conv(sd[p + ".conv1.weight"], x, t, stride, 1, act=1, bn=self._bn(p + ".bn1"), name=p + ".conv1")
        sc = x
        if (p + ".downsample.0.weight") in sd:
            sc = n.buffer(cout, ho, wo, 0)
            n.conv(sd[p + ".downsample.0.weight"], x, sc, stride, 0, act=0, bn=self._bn(p + ".downsample.1"), name=p + ".downsample")
        y = out if out is not None else n.buffer(cout, ho, wo, 1)
        n.conv(sd[p + ".conv2.weight"], t, y, 1, 1, act=1, bn=self._bn(p + ".bn2"), res_buf=sc, out_coff=out_coff, name=p + ".conv2")
        return y, ho, wo

    def _arm(self, n, p, x, cin, h, w):
        """AttentionRefinementModule (model.py:55-75) up to the attention vector: returns (feat, atten 1x1 map)"""
        sd = self._sd
        feat = n.buffer(128, h, w, 0)
        self._cbr(n, p + ".conv", x, feat, 3, 1, 1)
        pooled, att = n.buffer(128, 1, 1, 0), n.buffer(128, 1, 1, 0)
        n.global_avgpool(feat, 128, pooled)
        n.conv(sd[p + ".conv_atten.weight"], pooled, att, 1, 0, act=2, bn=self._bn(p + ".bn_atten"), name=p + ".conv_atten")   # sigmoid(bn(conv))
        return feat, att

    def _head(self, n, p, x, mid, h, w, in_coff=0):
        """BiSeNetOutput (model.py:33-45): ConvBNReLU 3x3 -> 1x1 conv to n_classes (rows padded to a multiple of 4)"""
        sd = self._sd
        t = n.buffer(mid, h, w, 0)
        self._cbr(n, p + ".conv", x, t, 3, 1, 1, in_coff=in_coff)
        wo = sd[p + ".conv_out.weight"]
        npad = (self.n_classes + 3) // 4 * 4
        wp = torch.zeros((npad,) + tuple(wo.shape[1:]))
        wp[: self.n_classes] = wo
        o = n.buffer(npad, h, w, 0)
        n.conv(wp, t, o, 1, 0, act=0, name=p + ".conv_out")
        return o

    def _build(self, H, W):
        sd = self._sd
        n = Net(self.max_batch, self.precision, self.device)
        g = dict(net=n)
        x = g["inp"] = n.buffer(3, H, W, 3)
        # Resnet18 (resnet.py:60-85)
        h, w = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        c1 = n.buffer(64, h, w, 1)
        n.conv(sd["cp.resnet.conv1.weight"], x, c1, 2, 3, act=1, bn=self._bn("cp.resnet.bn1"), name="cp.resnet.conv1")
        h, w = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        x = n.buffer(64, h, w, 1)
        n.maxpool(c1, x, 3, 2, 1)
        x, h, w = self._block(n, "cp.resnet.layer1.0", x, 64, 64, 1, h, w)
        x, h, w = self._block(n, "cp.resnet.layer1.1", x, 64, 64, 1, h, w)
        x, h, w = self._block(n, "cp.resnet.layer2.0", x, 64, 128, 2, h, w)
        h8, w8 = (h + 2 - 3) // 1 + 1, (w + 2 - 3) // 1 + 1
        cat = n.buffer(256, h8, w8, 1)                                  # torch.cat([fsp, fcp], dim=1) of the FFM (model.py:191): [feat8 | feat_cp8]
        x, h, w = self._block(n, "cp.resnet.layer2.1", x, 128, 128, 1, h, w, out=cat, out_coff=0)
        # (views of a wider buffer: the next block reads channels [0, 128) of `cat`)
        f16, h16, w16 = self._block(n, "cp.resnet.layer3.0", cat, 128, 256, 2, h8, w8)
        f16, h16, w16 = self._block(n, "cp.resnet.layer3.1", f16, 256, 256, 1, h16, w16)
        f32, h32, w32 = self._block(n, "cp.resnet.layer4.0", f16, 256, 512, 2, h16, w16)
        f32, h32, w32 = self._block(n, "cp.resnet.layer4.1", f32, 512, 512, 1, h32, w32)
        # ContextPath (model.py:95-115)
        pooled, avg = n.buffer(512, 1, 1, 0), n.buffer(128, 1, 1, 0)
        n.global_avgpool(f32, 512, pooled)
        self._cbr(n, "cp.conv_avg", pooled, avg, 1, 1, 0)
        feat32, att32 = self._arm(n, "cp.arm32", f32, 512, h32, w32)
        sum32 = n.buffer(128, h32, w32, 0)
        n.scale_add(feat32, 128, sum32, s_buf=att32, v_buf=avg)         # feat32_arm + nearest-upsampled avg (a 1x1 map: a broadcast)
        up32 = n.buffer(128, h16, w16, 1)
        n.upsample_nearest(sum32, up32)
        cp16 = n.buffer(128, h16, w16, 1)
        self._cbr(n, "cp.conv_head32", up32, cp16, 3, 1, 1)
        feat16, att16 = self._arm(n, "cp.arm16", f16, 256, h16, w16)
        sum16 = n.buffer(128, h16, w16, 0)
        n.scale_add(feat16, 128, sum16, s_buf=att16, t_buf=cp16)        # feat16_arm + feat32_up
        up16 = n.buffer(128, h8, w8, 1)
        n.upsample_nearest(sum16, up16)
        self._cbr(n, "cp.conv_head16", up16, cat, 3, 1, 1, out_coff=128)   # feat_cp8 -> channels [128, 256) of the FFM input
        # FeatureFusionModule (model.py:190-201)
        feat = n.buffer(256, h8, w8, 0)
        self._cbr(n, "ffm.convblk", cat, feat, 1, 1, 0)
        pooled2, a1, a2 = n.buffer(256, 1, 1, 0), n.buffer(64, 1, 1, 0), n.buffer(256, 1, 1, 0)
        n.global_avgpool(feat, 256, pooled2)
        n.conv(sd["ffm.conv1.weight"], pooled2, a1, 1, 0, act=1, name="ffm.conv1")
        n.conv(sd["ffm.conv2.weight"], a1, a2, 1, 0, act=2, name="ffm.conv2")
        fuse = n.buffer(256, h8, w8, 1)
        n.scale_add(feat, 256, fuse, s_buf=a2, t_buf=feat)              # feat * atten + feat
        g["out"] = self._head(n, "conv_out", fuse, 256, h8, w8)
        if self.aux:
            g["out16"] = self._head(n, "conv_out16", cat, 64, h8, w8, in_coff=128)   # feat_cp8 = channels [128, 256) of the FFM input
            g["out32"] = self._head(n, "conv_out32", cp16, 64, h16, w16)
        return g

    def __call__(self, x):
        if self._sd is None:
            raise RuntimeError("BiSeNet: load_state_dict first (face_parsing/__init__.py:22-26)")
        x = torch.as_tensor(x)
        B, Cn, H, W = x.shape
        g = self._nets.get((H, W))
        if g is None:
            g = self._nets[(H, W)] = self._build(H, W)
        n = g["net"]
        n.set_input(g["inp"], x)
        n.run(B)
        outs = [n.output_bilinear(g["out"], self.n_classes, B, H, W)]                 # F.interpolate(..., (H, W), bilinear, align_corners=True)
        if self.aux:
            outs += [n.output_bilinear(g["out16"], self.n_classes, B, H, W), n.output_bilinear(g["out32"], self.n_classes, B, H, W)]
        return tuple(outs)

    forward = __call__
