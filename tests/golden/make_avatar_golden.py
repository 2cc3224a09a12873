"""Generates tests/golden/avatar_golden.npz by running the REFERENCE's own S3FD and BiSeNet modules (build container only: imports
/root/reference).  The seeded weights come from mere_fusion_amd.weights (same seed -> same tensors on the GPU box); only inputs / outputs are
stored.  `torchvision` (imported but unused by face_parsing/model.py) is absent here and stubbed; `Resnet18.init_weight` (a torch.load of the
ImageNet file, resnet.py:87-93) is skipped -- the full state dict is loaded right after, as FaceParsing.model_init does.

    python tests/golden/make_avatar_golden.py
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
from mere_fusion_amd import weights as W   # noqa: E402

REF = "/root/reference/musetalk/utils"


def ref_s3fd():
    spec = importlib.util.spec_from_file_location("ref_net_s3fd", os.path.join(REF, "face_detection", "detection", "sfd", "net_s3fd.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.s3fd


def ref_bisenet():
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    pkg = types.ModuleType("ref_face_parsing")
    pkg.__path__ = [os.path.join(REF, "face_parsing")]           # the package's __init__ (cv2, PIL, torchvision.transforms) is not executed
    sys.modules["ref_face_parsing"] = pkg
    resnet = importlib.import_module("ref_face_parsing.resnet")
    resnet.Resnet18.init_weight = lambda self, path: None
    return importlib.import_module("ref_face_parsing.model").BiSeNet


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}
    rng = np.random.default_rng(0)
    # S3FD: two 80 x 112 "images" in the value range of detect() (BGR minus the channel means, sfd/detect.py:20)
    x = (rng.uniform(0, 255, (2, 3, 80, 112)) - np.array([104, 117, 123]).reshape(1, 3, 1, 1)).astype(np.float32)
    net = ref_s3fd()()
    net.load_state_dict(W.make_s3fd_state_dict(0), strict=True)
    net.eval()
    with torch.no_grad():
        olist = net(torch.from_numpy(x))
    out["s3fd_x"] = x
    for i, o in enumerate(olist):
        out[f"s3fd_out{i}"] = o.numpy()
    # BiSeNet: one normalised 64 x 96 crop (ToTensor + Normalize, face_parsing/__init__.py:29-33)
    xb = ((rng.uniform(0, 1, (1, 3, 64, 96)) - np.array([0.485, 0.456, 0.406]).reshape(1, 3, 1, 1)) / np.array([0.229, 0.224, 0.225]).reshape(1, 3, 1, 1)).astype(np.float32)
    bnet = ref_bisenet()("unused")
    bnet.load_state_dict(W.make_bisenet_state_dict(0), strict=True)
    bnet.eval()
    with torch.no_grad():
        f0, f16, f32 = bnet(torch.from_numpy(xb))
    out["bisenet_x"] = xb
    out["bisenet_out"] = f0.numpy()
    out["bisenet_out16_sum"] = np.array([f16.double().sum().item(), f16.double().abs().sum().item()])
    out["bisenet_out32_sum"] = np.array([f32.double().sum().item(), f32.double().abs().sum().item()])
    out["bisenet_out16_row"] = f16[0, :, 31, :].numpy()
    out["bisenet_out32_row"] = f32[0, :, 31, :].numpy()
    path = os.path.join(ROOT, "tests", "golden", "avatar_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", "s3fd outs", [tuple(o.shape) for o in olist], "bisenet", tuple(f0.shape),
          "argmax classes", np.unique(f0.numpy().argmax(1)).size, "s3fd cls1 range", float(olist[0].min()), float(olist[0].max()))


if __name__ == "__main__":
    main()
