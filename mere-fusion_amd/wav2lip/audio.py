"""`wav2lip.audio` drop-in, hot-path subset: `melspectrogram(wav)` (wav2lip/audio.py:45-51) on the GPU.

lipasr.py:23 calls `audio.melspectrogram(inputs)` with a float32 ndarray of (2B+l+r)*320 samples
and then indexes the result as `mel[:, a:b]` / `len(mel[0])` (lipasr.py:31-34), so a float64
ndarray of shape (80, T) comes back, exactly like the reference.
"""
import os

import numpy as np
import torch

from .. import ops

# wav2lip/hparams.py:33-73 (fixed at build time inside csrc/mf_mel.hip)
num_mels, n_fft, hop_size, win_size, sample_rate = 80, 800, 200, 800, 16000
preemphasis_k, min_level_db, ref_level_db, fmin, fmax, max_abs_value = 0.97, -100, 20, 55, 7600, 4.0

# librosa >= 0.10 pads the centred STFT with zeros ("constant"); older releases reflect.
PAD_MODES = {"constant": 0, "zeros": 0, "reflect": 1}
pad_mode = os.environ.get("MF_MEL_PAD_MODE", "constant")


def melspectrogram_device(wav, device=None):
    """torch fp32 [80, T] on the device (no host round trip)."""
    if not torch.is_tensor(wav):
        wav = torch.from_numpy(np.ascontiguousarray(wav, dtype=np.float32))
    if not wav.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("wav2lip.audio.melspectrogram needs a HIP device; no CPU path exists here")
        wav = wav.to(device or "cuda")
    return ops.melspec(wav.reshape(-1), PAD_MODES[pad_mode])


def melspectrogram(wav):
    return melspectrogram_device(wav).cpu().numpy().astype(np.float64)
