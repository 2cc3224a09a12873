"""TEST INFRASTRUCTURE (oracle/): CPU restatement of `BiSeNet.forward` (musetalk/utils/face_parsing/model.py:245-262) with its ContextPath
(:95-115), AttentionRefinementModule (:66-75), FeatureFusionModule (:190-201), BiSeNetOutput (:42-45) and Resnet18 / BasicBlock
(resnet.py:33-47,76-85) as a function of the state dict (eval-mode BatchNorm, eps 1e-5).  PINNED by tests/golden/avatar_golden.npz (the
reference's own module run in the build container).  Never imported by the product."""
import torch
import torch.nn.functional as F


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def _cbr(sd, p, x, stride=1, pad=1):                                    # ConvBNReLU, model.py:25-28
    return F.relu(_bn(sd, p + ".bn", F.conv2d(x, sd[p + ".conv.weight"], None, stride, pad)))


def _block(sd, p, x, stride):                                           # BasicBlock.forward, resnet.py:33-47
    r = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1)))
    r = _bn(sd, p + ".bn2", F.conv2d(r, sd[p + ".conv2.weight"], None, 1, 1))
    sc = x
    if p + ".downsample.0.weight" in sd:
        sc = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0))
    return F.relu(sc + r)


def _arm(sd, p, x):                                                     # AttentionRefinementModule.forward, model.py:66-75
    feat = _cbr(sd, p + ".conv", x)
    att = F.avg_pool2d(feat, feat.size()[2:])
    att = torch.sigmoid(_bn(sd, p + ".bn_atten", F.conv2d(att, sd[p + ".conv_atten.weight"])))
    return feat * att


def _out(sd, p, x):                                                     # BiSeNetOutput.forward, model.py:42-45
    return F.conv2d(_cbr(sd, p + ".conv", x), sd[p + ".conv_out.weight"])


def bisenet_forward(sd, x):
    H, W = x.shape[2:]
    h = F.relu(_bn(sd, "cp.resnet.bn1", F.conv2d(x, sd["cp.resnet.conv1.weight"], None, 2, 3)))       # resnet.py:77-79
    h = F.max_pool2d(h, 3, 2, 1)
    for b in (0, 1):
        h = _block(sd, f"cp.resnet.layer1.{b}", h, 1)
    feat8 = _block(sd, "cp.resnet.layer2.1", _block(sd, "cp.resnet.layer2.0", h, 2), 1)
    feat16 = _block(sd, "cp.resnet.layer3.1", _block(sd, "cp.resnet.layer3.0", feat8, 2), 1)
    feat32 = _block(sd, "cp.resnet.layer4.1", _block(sd, "cp.resnet.layer4.0", feat16, 2), 1)
    avg = _cbr(sd, "cp.conv_avg", F.avg_pool2d(feat32, feat32.size()[2:]), 1, 0)                       # model.py:103-105
    avg_up = F.interpolate(avg, feat32.shape[2:], mode="nearest")
    feat32_sum = _arm(sd, "cp.arm32", feat32) + avg_up
    feat32_up = _cbr(sd, "cp.conv_head32", F.interpolate(feat32_sum, feat16.shape[2:], mode="nearest"))
    feat16_sum = _arm(sd, "cp.arm16", feat16) + feat32_up
    feat16_up = _cbr(sd, "cp.conv_head16", F.interpolate(feat16_sum, feat8.shape[2:], mode="nearest"))
    fcat = torch.cat([feat8, feat16_up], dim=1)                                                        # FeatureFusionModule.forward, :190-201
    feat = _cbr(sd, "ffm.convblk", fcat, 1, 0)
    att = F.avg_pool2d(feat, feat.size()[2:])
    att = torch.sigmoid(F.conv2d(F.relu(F.conv2d(att, sd["ffm.conv1.weight"])), sd["ffm.conv2.weight"]))
    fuse = feat * att + feat
    outs = [_out(sd, "conv_out", fuse), _out(sd, "conv_out16", feat16_up), _out(sd, "conv_out32", feat32_up)]
    return [F.interpolate(o, (H, W), mode="bilinear", align_corners=True) for o in outs]                # :257-259
