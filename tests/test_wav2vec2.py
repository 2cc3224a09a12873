"""ER-NeRF audio front-end: the wav2vec2 CTC network NerfASR runs per step (nerfasr.py:38-45,105-143).

Oracle = the dependency itself: transformers' Wav2Vec2FeatureExtractor + Wav2Vec2ForCTC on the CPU in fp32 (oracle/wav2vec2_ref.py; the
library is installed here and on the GPU box, the reference never travels).  CPU tier: the seeded state dict loads into the library's model
with strict=True (key / shape manifest), parameter counts, the feature-ring glue against hand-derived indices.  GPU tier: logits of the
HIP stage vs the library, reduced and full XLSR-53-large size, several windows per call.  Tolerance: L-inf <= 2e-3 on logits of magnitude ~3
(bf16x3 products, fp32 accumulation, 24 pre-LN layers) -- the reference itself runs this model in fp32."""
import numpy as np
import pytest
import torch

from mere_fusion_amd import weights as W
from oracle import wav2vec2_ref as R


def test_state_dict_manifest_matches_transformers():
    from transformers import Wav2Vec2Config, Wav2Vec2ForCTC
    cfg = W.WAV2VEC2_SMALL
    sd = W.make_wav2vec2_state_dict(cfg, 0)
    model = R.build(cfg, sd)                                                  # strict=True inside
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == W.make_wav2vec2_state_dict(cfg, shapes_only=True)
    # the architecture app.py:660 names: XLSR-53 large + a 44-symbol head = 315,483,820 parameters (incl. the training-only masked_spec_embed)
    shapes = W.make_wav2vec2_state_dict(W.WAV2VEC2_XLSR_LARGE, shapes_only=True)
    assert sum(int(np.prod(s)) for s in shapes.values()) == 315_483_820
    assert shapes["lm_head.weight"] == (44, 1024)                             # nerfasr.py:19-20: audio_dim 44 for the esperanto model


def test_oracle_window_geometry():
    """(l + m + r) * 320 = 8960 samples -> 27 frames of 20 ms; logits[:, l : T - r + 1] keeps m = 8 rows (nerfasr.py:138-141)"""
    cfg = W.WAV2VEC2_SMALL
    model = R.build(cfg, W.make_wav2vec2_state_dict(cfg, 0))
    lg = R.frame_to_logits(model, W.make_speech_like_wav(8960, 0))
    assert lg.shape == (1, 27, 44)
    assert R.slice_logits(lg, 10, 10).shape == (1, 8, 44)
    # the processor's normalisation makes the result invariant to gain and offset of the waveform
    wav = W.make_speech_like_wav(8960, 1)
    a, b = R.frame_to_logits(model, wav), R.frame_to_logits(model, 0.25 * wav + 0.1)
    np.testing.assert_allclose(a, b, atol=2e-4)


class _StubModel:
    """returns logits whose row t carries the value 100 * call + t in every column"""
    def __init__(self):
        self.calls = 0

    def __call__(self, x):
        T = 27
        out = (100.0 * self.calls + torch.arange(T, dtype=torch.float32))[None, :, None].expand(1, T, 44).clone()
        self.calls += 1
        return type("R", (), {"logits": out})()


def test_frontend_ring_matches_hand_derived_indices():
    """nerfasr.py:48-58,75-124 with m = 8, l = r = 10: the first network call happens on the 18th run_step (10 zero frames are pre-loaded),
    consumes 28 frames and keeps logits rows 10..17 (values 10..17) in ring rows 0..7; afterwards one call per 8 steps.  get_next_feat's first
    window is ring rows [24..31, 0..7], then it advances by 2 rows per call."""
    from mere_fusion_amd.ernerf.asr import NerfASRFrontend
    stub = _StubModel()
    fe = NerfASRFrontend(stub, m=8, l=10, r=10, att=2, device="cpu")
    for i in range(17):
        fe.run_step()
    assert stub.calls == 0
    fe.run_step()
    assert stub.calls == 1 and len(fe.frames) == 20
    np.testing.assert_array_equal(fe.feat_queue[:8, 0].numpy(), np.arange(10, 18))
    assert (fe.feat_queue[8:] == 0).all()
    for i in range(8):
        fe.run_step()
    assert stub.calls == 2
    np.testing.assert_array_equal(fe.feat_queue[8:16, 3].numpy(), 100 + np.arange(10, 18))
    a = fe.get_next_feat()                                                    # [8, 44, 16]: 4 zero windows + 4 fresh ones
    assert a.shape == (8, 44, 16)
    assert (a[:4] == 0).all()
    np.testing.assert_array_equal(a[4, 0].numpy(), np.r_[np.zeros(8), np.arange(10, 18)])           # ring rows 24..31, 0..7
    np.testing.assert_array_equal(a[5, 0].numpy(), np.r_[np.zeros(6), np.arange(10, 18), [110, 111]])  # rows 26..31, 0..9
    b = fe.get_next_feat()
    np.testing.assert_array_equal(b[:7].numpy(), a[1:].numpy())              # the attention window slides by one
    np.testing.assert_array_equal(b[7, 0].numpy(), np.r_[np.arange(10, 18), 100 + np.arange(10, 18)])   # rows 0..15


def _run_hip(cfg, sd, wav, **kw):
    from mere_fusion_amd.ernerf.asr import HipWav2Vec2ForCTC, RawProcessor
    m = HipWav2Vec2ForCTC(cfg, sd, **kw)
    wav = np.atleast_2d(wav)
    return m(torch.from_numpy(wav)).logits.cpu().numpy(), m


@pytest.mark.gpu
@pytest.mark.parametrize("S", [1, 3])
def test_hip_wav2vec2_small_vs_transformers(lib_built, S):
    cfg = W.WAV2VEC2_SMALL
    sd = W.make_wav2vec2_state_dict(cfg, 0)
    wav = np.stack([W.make_speech_like_wav(8960, s) * (0.3 + 0.3 * s) for s in range(S)])
    want = R.frame_to_logits(R.build(cfg, sd), wav)
    got, _ = _run_hip(cfg, sd, wav, max_windows=S)
    assert got.shape == want.shape == (S, 27, 44)
    err = np.abs(got - want).max()
    print(f"[wav2vec2 small, {S} windows] logits L-inf vs transformers {err:.3e} (|logits| max {np.abs(want).max():.2f})")
    assert err <= 2e-3


@pytest.mark.gpu
def test_hip_wav2vec2_xlsr_large_vs_transformers(lib_built):
    """the full-size network nerfasr.py loads (24 layers x 1024, 315 M parameters), one (l + m + r) window, and the frontend's slice"""
    from mere_fusion_amd.ernerf.asr import NerfASRFrontend
    cfg = W.WAV2VEC2_XLSR_LARGE
    sd = W.make_wav2vec2_state_dict(cfg, 0)
    wav = W.make_speech_like_wav(8960, 5)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    want = R.frame_to_logits(R.build(cfg, sd), wav)
    got, m = _run_hip(cfg, sd, wav)
    err = np.abs(got - want).max()
    print(f"[wav2vec2 XLSR-53 large] logits L-inf vs transformers {err:.3e} (|logits| max {np.abs(want).max():.2f})")
    assert err <= 2e-3
    assert (got.argmax(-1) == want.argmax(-1)).mean() >= 0.96              # the CTC labels nerfasr would read off
    fe = NerfASRFrontend(m, m=8, l=10, r=10)
    feats = fe.frame_to_logits(wav)
    np.testing.assert_allclose(feats.cpu().numpy(), R.slice_logits(want, 10, 10)[0], atol=2e-3)
    # a second window length rebuilds the handle (warm-up / other l, m, r settings)
    wav2 = W.make_speech_like_wav(28 * 320 + 640, 6)
    got2 = m(torch.from_numpy(wav2)[None]).logits.cpu().numpy()
    want2 = R.frame_to_logits(R.build(cfg, sd), wav2)
    assert got2.shape == want2.shape and np.abs(got2 - want2).max() <= 2e-3


@pytest.mark.gpu
def test_hip_wav2vec2_batch_composition_and_errors(lib_built):
    cfg = W.WAV2VEC2_SMALL
    sd = W.make_wav2vec2_state_dict(cfg, 0)
    wav = np.stack([W.make_speech_like_wav(8960, s) for s in range(4)])
    got4, m = _run_hip(cfg, sd, wav, max_windows=4)
    got1 = m(torch.from_numpy(wav[2:3])).logits.cpu().numpy()
    np.testing.assert_allclose(got4[2:3], got1, atol=2e-5)                   # a window's logits do not depend on its neighbours
    with pytest.raises(RuntimeError, match="capacity"):
        m(torch.from_numpy(np.concatenate([wav, wav])))
    bad = dict(cfg, feat_extract_norm="group")
    with pytest.raises(ValueError, match="layer"):
        _run_hip(bad, sd, wav[:1])
    sd2 = {k: v for k, v in sd.items() if "layers.1.attention.k_proj.weight" not in k}
    with pytest.raises(RuntimeError, match="k_proj.weight"):
        _run_hip(cfg, sd2, wav[:1])
