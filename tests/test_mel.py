"""Wav2Lip mel-spectrogram (H1): oracle known-answer tests on CPU, HIP kernel vs oracle on GPU.
The reference holds no vectors for this function and librosa is absent: parity is unpinned at the
librosa boundary (oracle/mel_ref.py header); these are self-consistency KATs."""
import numpy as np
import pytest
import torch

from oracle import mel_ref


def _wav(n, seed=0):
    rng = np.random.default_rng(seed)
    return (0.1 * rng.standard_normal(n)).clip(-1, 1).astype(np.float32)


def test_oracle_shape_and_range():
    for n, T in [(16640, 84), (7040, 36), (48000, 241)]:
        m = mel_ref.melspectrogram(_wav(n))
        assert m.shape == (80, T) and m.dtype == np.float64
        assert m.min() >= -4 and m.max() <= 4


def test_oracle_silence_is_floor():
    assert (mel_ref.melspectrogram(np.zeros(7040, np.float32)) == -4.0).all()


def test_oracle_tone_peaks_in_its_band():
    t = np.arange(16000) / 16000
    edges = mel_ref.mel_to_hz(np.linspace(mel_ref.hz_to_mel(55), mel_ref.hz_to_mel(7600), 82))
    for f in (440.0, 1000.0, 3000.0):
        m = mel_ref.melspectrogram((0.5 * np.sin(2 * np.pi * f * t)).astype(np.float32))
        b = int(m[:, 40].argmax())
        assert edges[b] <= f <= edges[b + 2], (f, b)


def test_oracle_mel_basis_properties():
    B = mel_ref.mel_basis()
    assert B.shape == (80, 401) and B.dtype == np.float32 and (B >= 0).all()
    # Slaney area normalisation: each triangle integrates to ~1 over Hz (bin width 20 Hz)
    np.testing.assert_allclose(B.sum(1) * 20.0, 1.0, atol=0.12)
    assert B[:, 0].sum() == 0            # DC is below fmin=55 Hz
    assert B[:, 381:].sum() == 0         # above fmax=7600 Hz (bin 380)


def test_streaming_windows_ignore_pad_mode():
    # only the first/last 2 frames see the padding; chunks use frames 16..79
    w = _wav(16640, 3)
    a = mel_ref.melspectrogram(w, "constant")
    b = mel_ref.melspectrogram(w, "reflect")
    np.testing.assert_array_equal(a[:, 2:82], b[:, 2:82])
    assert np.abs(a[:, :2] - b[:, :2]).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("n", [16640, 7040, 48000, 520, 200])
@pytest.mark.parametrize("mode", ["constant", "reflect"])
def test_hip_mel_matches_oracle(lib_built, n, mode):
    from mere_fusion_amd import ops
    from mere_fusion_amd.wav2lip import audio
    if mode == "reflect" and n <= 400:
        pytest.skip("np.pad reflect needs n > n_fft/2")
    w = _wav(n, n)
    want = mel_ref.melspectrogram(w, mode)
    got = ops.melspec(torch.from_numpy(w).cuda(), audio.PAD_MODES[mode]).cpu().numpy()
    assert got.shape == want.shape
    # fp64 DFT on both sides, fp32 output: tolerance 1e-5 on the [-4, 4] scale
    assert np.abs(got - want).max() <= 1e-5


@pytest.mark.gpu
def test_hip_mel_kats(lib_built):
    from mere_fusion_amd.wav2lip import audio
    m = audio.melspectrogram(np.zeros(7040, np.float32))
    assert m.shape == (80, 36) and m.dtype == np.float64 and (m == -4.0).all()
    t = np.arange(16000) / 16000
    m = audio.melspectrogram((0.5 * np.sin(2 * np.pi * 440 * t)).astype(np.float32))
    assert int(m[:, 40].argmax()) == 10
    with pytest.raises(RuntimeError, match="empty"):
        audio.melspectrogram(np.zeros(0, np.float32))
