"""ORACLE (test infrastructure, not product): numpy/scipy restatement of the Wav2Lip mel-spectrogram.

Follows wav2lip/audio.py:20-23 (preemphasis), :45-51 (melspectrogram), :57-61 (_stft),
:92-101 (_linear_to_mel, _build_mel_basis), :103-105 (_amp_to_db), :110-116 (_normalize) with the
constants of wav2lip/hparams.py:33-73.

PARITY UNPINNED at the librosa boundary: the reference calls `librosa.stft` and
`librosa.filters.mel` (audio.py:61,100); librosa is an un-vendored, unpinned dependency
(requirements.txt:6) that is absent from the build container, and the reference holds no tests
or vectors for this function.  What is restated here is librosa's published algorithm:
  * stft(y, n_fft=800, hop_length=200, win_length=800): center=True, window =
    scipy.signal.get_window('hann', 800, fftbins=True), pad_mode="constant" (zeros) for
    librosa >= 0.10 -- the keyword-only call style at audio.py:61 needs >= 0.10 -- and
    "reflect" for older releases; both are exposed.
  * filters.mel(sr, n_fft, n_mels, fmin, fmax): Slaney mel scale (htk=False), Slaney area
    normalisation, float32 result.
Known-answer checks in tests/test_mel.py: silence -> -4 everywhere; a pure tone peaks in the
mel band containing its frequency; frame count T = 1 + n//200; the steady-state windows the
streaming loop consumes (frames 16..79 of 84) do not depend on pad_mode.

Still unpinned against librosa itself, but no longer single-source: tests/test_mel.py also holds this file
against two independent implementations present in the image --
  * `torch.stft` (centred, periodic Hann, constant / reflect padding): the STFT to 1e-10;
  * `transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney")` -- a published
    re-implementation of `librosa.filters.mel` -- for the (80, 401) basis to 2e-7, and
    `transformers.audio_utils.spectrogram` for the whole magnitude -> mel -> dB chain to 1e-5.
"""
import numpy as np
from scipy import signal

N_FFT, HOP, WIN, SR, N_MELS = 800, 200, 800, 16000, 80
FMIN, FMAX = 55, 7600
PREEMPH, MIN_LEVEL_DB, REF_LEVEL_DB, MAX_ABS = 0.97, -100, 20, 4.0


def hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_basis():
    """librosa.filters.mel(sr=16000., n_fft=800, n_mels=80, fmin=55, fmax=7600) -> (80, 401) float32."""
    weights = np.zeros((N_MELS, 1 + N_FFT // 2), dtype=np.float32)
    fftfreqs = np.linspace(0, SR / 2.0, 1 + N_FFT // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(FMIN), hz_to_mel(FMAX), N_MELS + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(N_MELS):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:N_MELS + 2] - mel_f[:N_MELS])
    weights *= enorm[:, np.newaxis]
    return weights


def stft(y, pad_mode="constant"):
    """Centred STFT magnitude-ready complex matrix (401, T), T = 1 + len(y)//200."""
    y = np.asarray(y, dtype=np.float64)
    win = signal.get_window("hann", WIN, fftbins=True)
    yp = np.pad(y, N_FFT // 2, mode=pad_mode)
    T = 1 + len(y) // HOP
    frames = np.stack([yp[t * HOP: t * HOP + N_FFT] * win for t in range(T)], axis=1)   # (800, T)
    return np.fft.rfft(frames, n=N_FFT, axis=0)


_basis = None


def melspectrogram(wav, pad_mode="constant"):
    global _basis
    if _basis is None:
        _basis = mel_basis()
    y = signal.lfilter([1, -PREEMPH], [1], wav)                       # float64
    D = stft(y, pad_mode)
    mel = np.dot(_basis, np.abs(D))
    min_level = np.exp(MIN_LEVEL_DB / 20 * np.log(10))
    S = 20 * np.log10(np.maximum(min_level, mel)) - REF_LEVEL_DB
    return np.clip((2 * MAX_ABS) * ((S - MIN_LEVEL_DB) / (-MIN_LEVEL_DB)) - MAX_ABS, -MAX_ABS, MAX_ABS)
