"""The 11 distinct conv geometries of the Wav2Lip generator as small seeded
cases: same kernel / stride / padding / residual structure as the real layers, spatial sizes of the
real layers, channel counts kept modest so the fixtures stay small.  Parameters and inputs are
regenerated from the seed (numpy PCG64); only the reference's OUTPUTS are stored in
tests/golden/conv_golden.npz."""
import numpy as np
import torch

TAP_STRIDE = 97     # sampling of the per-block activations kept in wav2lip_golden.npz
TAP_MAX = 4096

CASES = [
    dict(name="k7_s1_p3_cin6", cin=6, cout=16, k=7, stride=1, pad=3, transposed=0, outpad=0, residual=0, h=48, w=48),
    dict(name="k3_s1_p1", cin=32, cout=64, k=3, stride=1, pad=1, transposed=0, outpad=0, residual=0, h=24, w=24),
    dict(name="k3_s1_p1_res", cin=64, cout=64, k=3, stride=1, pad=1, transposed=0, outpad=0, residual=1, h=24, w=24),
    dict(name="k3_s2_p1", cin=16, cout=32, k=3, stride=2, pad=1, transposed=0, outpad=0, residual=0, h=48, w=48),
    dict(name="k3_s31_p1", cin=32, cout=64, k=3, stride=(3, 1), pad=1, transposed=0, outpad=0, residual=0, h=80, w=16),
    dict(name="k3_s3_p1", cin=64, cout=128, k=3, stride=3, pad=1, transposed=0, outpad=0, residual=0, h=27, w=16),
    dict(name="k3_s32_p1", cin=128, cout=256, k=3, stride=(3, 2), pad=1, transposed=0, outpad=0, residual=0, h=9, w=6),
    dict(name="k3_s1_p0", cin=256, cout=512, k=3, stride=1, pad=0, transposed=0, outpad=0, residual=0, h=3, w=3),
    dict(name="k1_s1_p0", cin=512, cout=512, k=1, stride=1, pad=0, transposed=0, outpad=0, residual=0, h=1, w=1),
    dict(name="k1_cin1", cin=1, cout=32, k=3, stride=1, pad=1, transposed=0, outpad=0, residual=0, h=80, w=16),
    dict(name="convT_s1_p0_1x1", cin=1024, cout=512, k=3, stride=1, pad=0, transposed=1, outpad=0, residual=0, h=1, w=1),
    dict(name="convT_s2_p1_op1", cin=160, cout=64, k=3, stride=2, pad=1, transposed=1, outpad=1, residual=0, h=12, w=12),
    dict(name="convT_s2_p1_op1_wide", cin=768, cout=384, k=3, stride=2, pad=1, transposed=1, outpad=1, residual=0, h=6, w=6),
    dict(name="k3_s1_p1_cin80", cin=80, cout=32, k=3, stride=1, pad=1, transposed=0, outpad=0, residual=0, h=24, w=24),
]
BATCH = 2


def _rng(case, salt=0):
    return np.random.default_rng(sum(map(ord, case["name"])) * 7919 + salt)


def case_params(case):
    rng = _rng(case)
    ci, co, k = case["cin"], case["cout"], case["k"]
    shape = (ci, co, k, k) if case["transposed"] else (co, ci, k, k)
    f = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    return dict(
        weight=f(rng.standard_normal(shape) * np.sqrt(2.0 / (ci * k * k))),
        bias=f(rng.standard_normal(co) * 0.1),
        gamma=f(rng.uniform(0.5, 1.5, co)), beta=f(rng.standard_normal(co) * 0.2),
        mean=f(rng.standard_normal(co) * 0.2), var=f(rng.uniform(0.5, 1.5, co)))


def case_input(case):
    rng = _rng(case, salt=1)
    x = rng.standard_normal((BATCH, case["cin"], case["h"], case["w"])).astype(np.float32)
    return torch.from_numpy(np.maximum(x, -0.5))   # mostly post-ReLU-like, a few negatives
