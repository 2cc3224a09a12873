"""Fused attention kernel (mf_attn.hip) vs the oracle's attention core, through the C ABI.

The oracle core is oracle/musetalk_ref.py::attention_core (softmax(q k^T / sqrt(dh)) v per head, the arithmetic of the
diffusers Attention that musetalk/models/unet.py:36-47 instantiates and of whisper/model.py:82-93)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import musetalk_ref as R


def _qkv(b, tq, tk, heads, dh, seed):
    g = torch.Generator().manual_seed(seed)
    c = heads * dh
    # non-symmetric, per-channel scaled inputs: an operand transpose or a head / channel mix-up changes the answer
    ramp = torch.linspace(0.5, 1.5, c)
    q = torch.randn(b, tq, c, generator=g) * ramp
    k = torch.randn(b, tk, c, generator=g) * ramp.flip(0)
    v = torch.randn(b, tk, c, generator=g) + torch.arange(c) * 0.01
    return q, k, v


def test_oracle_attention_core_matches_dense_softmax():
    q, k, v = _qkv(2, 5, 7, 2, 8, 0)
    got = R.attention_core(q, k, v, 2)
    qh = q.view(2, 5, 2, 8).permute(0, 2, 1, 3).double()
    kh = k.view(2, 7, 2, 8).permute(0, 2, 1, 3).double()
    vh = v.view(2, 7, 2, 8).permute(0, 2, 1, 3).double()
    w = torch.softmax(qh @ kh.transpose(-1, -2) / np.sqrt(8.0), -1)
    want = (w @ vh).permute(0, 2, 1, 3).reshape(2, 5, 16)
    assert (got.double() - want).abs().max() < 1e-6


def _hip_attention(q, k, v, heads, precision):
    from mere_fusion_amd import _lib
    L = _lib.lib()
    _lib.init_device(0)
    b, tq, c = q.shape
    tk = k.shape[1]
    qd, kd, vd = (t.contiguous().cuda() for t in (q, k, v))
    out = torch.empty(b, tq, c, device="cuda")
    _lib.check(L.mf_attention_forward(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), out.data_ptr(), b, tq, tk, heads, c // heads,
                                      _lib.PRECISIONS[precision], None))
    return out.cpu()


# (batch, tq, tk, heads, dh): the UNet's self / cross shapes, Whisper's, ragged tails, fewer queries than one wave
CASES = [
    (2, 1024, 1024, 8, 40), (2, 1024, 50, 8, 40), (2, 256, 256, 8, 80), (2, 256, 50, 8, 80),
    (2, 64, 64, 8, 160), (2, 64, 50, 8, 160), (1, 16, 16, 8, 160), (1, 16, 50, 8, 160),
    (1, 1500, 1500, 6, 64), (1, 100, 37, 2, 40), (3, 70, 129, 1, 80), (1, 1, 1, 1, 64), (8, 1024, 1024, 8, 40),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda c: "b%d_q%d_k%d_h%d_d%d" % c)
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
def test_hip_attention_matches_oracle(lib_built, case, precision):
    b, tq, tk, heads, dh = case
    q, k, v = _qkv(b, tq, tk, heads, dh, tq * 7 + tk)
    want = R.attention_core(q, k, v, heads)
    got = _hip_attention(q, k, v, heads, precision)
    err = (got - want).abs().max().item()
    # outputs are convex combinations of v (|v| <~ 8): x3 carries ~16 mantissa bits end to end, bf16 8
    tol = 2e-4 if precision == "bf16x3" else 8e-2
    assert err <= tol, f"L_inf {err:.3e} > {tol}"


@pytest.mark.gpu
def test_hip_attention_rejects_unsupported_head_dim(lib_built):
    q, k, v = _qkv(1, 8, 8, 1, 24, 0)
    with pytest.raises(RuntimeError, match="head_dim"):
        _hip_attention(q, k, v, 1, "bf16x3")
