"""Avatar preparation networks: S3FD face detector (net_s3fd.py) and BiSeNet face parser (face_parsing/model.py).

Golden = the reference's own modules run in the build container on seeded weights / inputs (tests/golden/make_avatar_golden.py ->
avatar_golden.npz).  CPU tier: the oracle restatements (oracle/s3fd_ref.py, oracle/bisenet_ref.py) against the golden; state-dict manifests.
GPU tier: the drop-in modules (mere-fusion_amd/avatar) against the golden and, at other sizes / batches, against the oracle.
Tolerances (bf16x3 products, fp32 accumulation): S3FD head outputs 1e-3 absolute (levels on L2-normalised features) / 5e-3 (levels on raw
activations of magnitude ~1e2) on values of magnitude ~0.5, 1.5e-3 on the softmax scores detect() thresholds; BiSeNet logits 2e-3 x max|logit| and identical argmax on >= 99.9 % of the pixels."""
import os

import numpy as np
import pytest
import torch

from mere_fusion_amd import weights as W
from oracle import bisenet_ref, s3fd_ref

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "avatar_golden.npz"))


def test_oracle_s3fd_matches_reference_golden(golden):
    sd = W.make_s3fd_state_dict(0)
    with torch.no_grad():
        outs = s3fd_ref.s3fd_forward(sd, torch.from_numpy(golden["s3fd_x"]))
    assert len(outs) == 12
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.numpy(), golden[f"s3fd_out{i}"], rtol=0, atol=2e-5, err_msg=f"output {i}")
    assert outs[0].shape[1] == 2 and outs[1].shape[1] == 4                         # max-out leaves (background, face)


def test_oracle_bisenet_matches_reference_golden(golden):
    sd = W.make_bisenet_state_dict(0)
    with torch.no_grad():
        f0, f16, f32 = bisenet_ref.bisenet_forward(sd, torch.from_numpy(golden["bisenet_x"]))
    scale = np.abs(golden["bisenet_out"]).max()
    np.testing.assert_allclose(f0.numpy(), golden["bisenet_out"], rtol=0, atol=2e-5 * scale)
    np.testing.assert_allclose(f16[0, :, 31, :].numpy(), golden["bisenet_out16_row"], rtol=0, atol=2e-5 * scale)
    np.testing.assert_allclose(f32[0, :, 31, :].numpy(), golden["bisenet_out32_row"], rtol=0, atol=2e-5 * scale)
    np.testing.assert_allclose(f16.double().abs().sum().item(), golden["bisenet_out16_sum"][1], rtol=1e-5)
    np.testing.assert_allclose(f32.double().abs().sum().item(), golden["bisenet_out32_sum"][1], rtol=1e-5)


def test_manifests():
    s = W.make_s3fd_state_dict(shapes_only=True)
    assert len(s) == 2 * 31 + 3 and s["fc6.weight"] == (1024, 512, 3, 3) and s["conv3_3_norm_mbox_conf.weight"] == (4, 256, 3, 3)
    assert sum(int(np.prod(v)) for v in s.values()) == 22_459_110                 # VGG16 trunk + extras + heads of net_s3fd.py:25-70
    b = W.make_bisenet_state_dict(shapes_only=True)
    assert b["cp.resnet.layer2.0.downsample.0.weight"] == (128, 64, 1, 1) and "cp.resnet.layer1.0.downsample.0.weight" not in b
    assert b["conv_out.conv_out.weight"] == (19, 256, 1, 1) and b["ffm.conv1.weight"] == (64, 256, 1, 1)


def _s3fd(**kw):
    from mere_fusion_amd.avatar import s3fd
    net = s3fd(**kw)
    net.load_state_dict(W.make_s3fd_state_dict(0))
    return net.to("cuda").eval()


@pytest.mark.gpu
def test_hip_s3fd_matches_reference_golden(lib_built, golden):
    outs = _s3fd()(torch.from_numpy(golden["s3fd_x"]))
    assert len(outs) == 12
    errs = []
    for i, o in enumerate(outs):
        want = golden[f"s3fd_out{i}"]
        assert tuple(o.shape) == want.shape
        errs.append(float(np.abs(o.cpu().numpy() - want).max()))
    print("[s3fd vs reference golden] max |diff| per output (cls1, reg1, ..., cls6, reg6): " + " ".join(f"{e:.1e}" for e in errs))
    # levels 1-3 read L2-normalised features (scale-free); levels 4-6 read raw VGG activations, which reach the hundreds for pixel-valued
    # inputs: an operand error of 2^-17 relative shows up as ~1e-5 x |activation| on heads whose own outputs are O(0.5)
    assert max(errs[:6]) <= 1e-3 and max(errs[6:]) <= 5e-3, errs
    # what detect() thresholds (sfd/detect.py:37-47): softmax face scores
    for i in range(6):
        got = torch.softmax(outs[2 * i], 1)[:, 1].cpu().numpy()
        want = torch.softmax(torch.from_numpy(golden[f"s3fd_out{2 * i}"]), 1)[:, 1].numpy()
        assert np.abs(got - want).max() <= 1.5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W_", [(1, 240, 320), (3, 96, 96), (1, 101, 135)])
def test_hip_s3fd_matches_oracle_other_sizes(lib_built, B, H, W_):
    rng = np.random.default_rng(B * 1000 + H)
    x = (rng.uniform(0, 255, (B, 3, H, W_)) - np.array([104, 117, 123]).reshape(1, 3, 1, 1)).astype(np.float32)
    sd = W.make_s3fd_state_dict(0)
    with torch.no_grad():
        want = s3fd_ref.s3fd_forward(sd, torch.from_numpy(x))
    net = _s3fd(max_batch=4)
    got = net(torch.from_numpy(x))
    for i, (g, w) in enumerate(zip(got, want)):
        assert tuple(g.shape) == tuple(w.shape), f"output {i}"
        assert np.abs(g.cpu().numpy() - w.numpy()).max() <= (1e-3 if i < 6 else 5e-3), f"output {i}"
    got2 = net(torch.from_numpy(x).cuda())                                          # graph replay, device input
    for g, g2 in zip(got, got2):
        assert torch.equal(g, g2)


def _bisenet(**kw):
    from mere_fusion_amd.avatar import BiSeNet
    net = BiSeNet("unused", **kw)
    net.cuda()
    net.load_state_dict(W.make_bisenet_state_dict(0))
    return net.eval()


@pytest.mark.gpu
def test_hip_bisenet_matches_reference_golden(lib_built, golden):
    got = _bisenet(aux=True)(torch.from_numpy(golden["bisenet_x"]))
    want = golden["bisenet_out"]
    scale = np.abs(want).max()
    g0 = got[0].cpu().numpy()
    err = np.abs(g0 - want).max()
    print(f"[bisenet vs reference golden] logits L-inf {err:.3e} (max |logit| {scale:.2f}), argmax agreement {(g0.argmax(1) == want.argmax(1)).mean():.5f}")
    assert err <= 2e-3 * scale
    assert (g0.argmax(1) == want.argmax(1)).mean() >= 0.999
    np.testing.assert_allclose(got[1][0, :, 31, :].cpu().numpy(), golden["bisenet_out16_row"], rtol=0, atol=2e-3 * scale)
    np.testing.assert_allclose(got[2][0, :, 31, :].cpu().numpy(), golden["bisenet_out32_row"], rtol=0, atol=2e-3 * scale)


@pytest.mark.gpu
def test_hip_bisenet_full_size_matches_oracle(lib_built):
    """the size FaceParsing uses (512 x 512, face_parsing/__init__.py:41-47) and what it reads off: argmax over the 19 classes"""
    rng = np.random.default_rng(5)
    x = ((rng.uniform(0, 1, (1, 3, 512, 512)) - np.array([0.485, 0.456, 0.406]).reshape(1, 3, 1, 1)) / np.array([0.229, 0.224, 0.225]).reshape(1, 3, 1, 1)).astype(np.float32)
    sd = W.make_bisenet_state_dict(0)
    with torch.no_grad():
        want = bisenet_ref.bisenet_forward(sd, torch.from_numpy(x))[0].numpy()
    net = _bisenet()
    got = net(torch.from_numpy(x))[0].cpu().numpy()
    scale = np.abs(want).max()
    agree = (got.argmax(1) == want.argmax(1)).mean()
    print(f"[bisenet 512x512 vs oracle] logits L-inf {np.abs(got - want).max():.3e} (max |logit| {scale:.2f}), argmax agreement {agree:.6f}")
    assert np.abs(got - want).max() <= 2e-3 * scale and agree >= 0.999
    parsing = got[0].argmax(0)                                                        # face_parsing/__init__.py:52-54
    parsing[parsing > 13] = 0
    parsing[parsing >= 1] = 255
    want_p = want[0].argmax(0); want_p[want_p > 13] = 0; want_p[want_p >= 1] = 255
    assert (parsing == want_p).mean() >= 0.999
    assert torch.equal(net(torch.from_numpy(x))[0], net(torch.from_numpy(x))[0])      # graph replay is deterministic
