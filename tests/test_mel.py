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


# ---- cross-checks against two INDEPENDENT implementations available in the image (the oracle is no longer single-source) ----
def test_oracle_stft_matches_torch_stft():
    """torch.stft with librosa's defaults -- centred, periodic Hann, zero ("constant") or reflect padding -- is an independent
    implementation of the transform `librosa.stft(y, n_fft=800, hop_length=200, win_length=800)` names (wav2lip/audio.py:61)."""
    y = torch.from_numpy(_wav(16640, 7)).double()
    for mode in ("constant", "reflect"):
        want = torch.stft(y, n_fft=800, hop_length=200, win_length=800, window=torch.hann_window(800, periodic=True, dtype=torch.float64),
                          center=True, pad_mode=mode, return_complex=True).numpy()
        got = mel_ref.stft(y.numpy(), mode)
        assert got.shape == want.shape == (401, 84)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-10)


def test_oracle_mel_basis_matches_transformers_filter_bank():
    """`transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney")` is a published re-implementation of
    `librosa.filters.mel` (its docstring says so); it returns the transpose (401, 80)."""
    from transformers.audio_utils import mel_filter_bank
    want = mel_filter_bank(num_frequency_bins=401, num_mel_filters=80, min_frequency=55, max_frequency=7600, sampling_rate=16000,
                           norm="slaney", mel_scale="slaney").T
    got = mel_ref.mel_basis()
    assert got.shape == want.shape == (80, 401)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-7)


def test_oracle_melspectrogram_matches_transformers_spectrogram():
    """The whole chain after the pre-emphasis through transformers' `spectrogram` (magnitude STFT -> mel -> 20 log10 with a floor), then
    the reference's own affine normalisation (wav2lip/audio.py:103-116)."""
    from scipy import signal
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
    wav = _wav(16640, 11)
    y = signal.lfilter([1, -0.97], [1], wav)
    fb = mel_filter_bank(num_frequency_bins=401, num_mel_filters=80, min_frequency=55, max_frequency=7600, sampling_rate=16000,
                         norm="slaney", mel_scale="slaney")
    mel = spectrogram(y, window_function(800, "hann", periodic=True), frame_length=800, hop_length=200, fft_length=800, power=1.0, center=True,
                      pad_mode="constant", mel_filters=fb, mel_floor=1e-5, dtype=np.float64)           # (80, 84) magnitudes, floored at 10^(-100/20)
    S = 20 * np.log10(mel) - 20
    want = np.clip(8 * ((S + 100) / 100) - 4, -4, 4)
    got = mel_ref.melspectrogram(wav, "constant")
    assert got.shape == want.shape == (80, 84)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-5)
