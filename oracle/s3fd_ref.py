"""TEST INFRASTRUCTURE (oracle/): CPU restatement of `s3fd.forward` (face_detection/detection/sfd/net_s3fd.py:72-129) and `L2Norm.forward`
(:15-19) as a function of the state dict.  PINNED: tests/golden/make_avatar_golden.py runs the reference's own module on seeded inputs in the
build container (tests/golden/avatar_golden.npz); tests/test_avatar.py holds this restatement to it.  Never imported by the product."""
import torch
import torch.nn.functional as F


def _l2norm(x, weight, eps=1e-10):                                    # net_s3fd.py:15-19
    norm = x.pow(2).sum(dim=1, keepdim=True).sqrt() + eps
    return x / norm * weight.view(1, -1, 1, 1)


def s3fd_forward(sd, x):
    def c(name, h, stride=1, pad=1):
        return F.conv2d(h, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=pad)

    h = F.relu(c("conv1_1", x)); h = F.relu(c("conv1_2", h)); h = F.max_pool2d(h, 2, 2)          # :73-75
    h = F.relu(c("conv2_1", h)); h = F.relu(c("conv2_2", h)); h = F.max_pool2d(h, 2, 2)
    h = F.relu(c("conv3_1", h)); h = F.relu(c("conv3_2", h)); h = F.relu(c("conv3_3", h)); f3_3 = h; h = F.max_pool2d(h, 2, 2)
    h = F.relu(c("conv4_1", h)); h = F.relu(c("conv4_2", h)); h = F.relu(c("conv4_3", h)); f4_3 = h; h = F.max_pool2d(h, 2, 2)
    h = F.relu(c("conv5_1", h)); h = F.relu(c("conv5_2", h)); h = F.relu(c("conv5_3", h)); f5_3 = h; h = F.max_pool2d(h, 2, 2)
    h = F.relu(c("fc6", h, 1, 3)); h = F.relu(c("fc7", h, 1, 0)); ffc7 = h                      # :98-100
    h = F.relu(c("conv6_1", h, 1, 0)); h = F.relu(c("conv6_2", h, 2, 1)); f6_2 = h
    h = F.relu(c("conv7_1", h, 1, 0)); h = F.relu(c("conv7_2", h, 2, 1)); f7_2 = h
    f3_3 = _l2norm(f3_3, sd["conv3_3_norm.weight"]); f4_3 = _l2norm(f4_3, sd["conv4_3_norm.weight"]); f5_3 = _l2norm(f5_3, sd["conv5_3_norm.weight"])
    outs = []
    for key, f in (("conv3_3_norm", f3_3), ("conv4_3_norm", f4_3), ("conv5_3_norm", f5_3), ("fc7", ffc7), ("conv6_2", f6_2), ("conv7_2", f7_2)):
        outs += [c(key + "_mbox_conf", f), c(key + "_mbox_loc", f)]
    chunk = torch.chunk(outs[0], 4, 1)                                                            # :123-126 max-out background label
    outs[0] = torch.cat([torch.max(torch.max(chunk[0], chunk[1]), chunk[2]), chunk[3]], dim=1)
    return outs
