"""Records tests/golden/whisper_golden.npz from the REAL reference Whisper code (build container only).

    python tests/golden/make_whisper_golden.py          # needs /root/reference

The reference modules are imported with stub `ffmpeg` / `soundfile` modules (only load_audio(file) and an
unused import touch them: whisper/audio.py:41-45, audio2feature.py:3).  Recorded (data only):
  mel_filters        the values of assets/mel_filters.npz["mel_80"] (the expected output of the filterbank
                     construction the docstring at audio.py:80-87 defines)
  logmel_16640       log_mel_spectrogram of the seeded 16640-sample signal           (80, 104)
  feat_16640/11520   Audio2Feature.audio2feat with a seeded tiny encoder, subsampled + float64 sums
  chunk_idx_*        the row indices feature2chunks selects (museasr.py:27 arguments)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
for name in ("ffmpeg", "soundfile"):
    sys.modules[name] = types.ModuleType(name)

from mere_fusion_amd import weights as W                                   # noqa: E402
from musetalk.whisper.whisper.model import Whisper, ModelDimensions        # noqa: E402
from musetalk.whisper.whisper import audio as ref_audio                    # noqa: E402
from musetalk.whisper import audio2feature as ref_a2f                      # noqa: E402

torch.set_num_threads(8)
FEAT_STRIDE = 13


def main():
    d = W.WHISPER_TINY
    dims = ModelDimensions(d["n_mels"], d["n_audio_ctx"], d["n_audio_state"], d["n_audio_head"], d["n_audio_layer"],
                           51865, 448, 384, 6, 4)
    model = Whisper(dims)
    sd = W.make_whisper_encoder_state_dict(0)
    missing, unexpected = model.encoder.load_state_dict(sd, strict=False)
    assert missing == ["positional_embedding"] and not unexpected, (missing, unexpected)
    model.eval()
    a2f = ref_a2f.Audio2Feature.__new__(ref_a2f.Audio2Feature)   # bypass load_model(path): no checkpoint ships
    a2f.model = model
    out = {"mel_filters": np.load(os.path.join(os.path.dirname(ref_audio.__file__), "assets", "mel_filters.npz"))["mel_80"]}
    for n, B in ((16640, 16), (11520, 8)):
        wav = W.make_speech_like_wav(n, 0)
        out[f"logmel_{n}"] = ref_audio.log_mel_spectrogram(wav).numpy()
        with torch.no_grad():
            feat = a2f.audio2feat(wav)
        print(n, "feat", feat.shape, float(np.abs(feat).max()), float(feat.std()))
        out[f"feat_shape_{n}"] = np.asarray(feat.shape)
        out[f"feat_sample_{n}"] = feat.reshape(-1)[::FEAT_STRIDE].copy()
        out[f"feat_abssum_{n}"] = np.float64(np.abs(feat.astype(np.float64)).sum())
        out[f"feat_first_{n}"] = feat[:2].copy()
        chunks = a2f.feature2chunks(feature_array=feat, fps=50 / 2, batch_size=B, start=10 / 2)
        idxs = [a2f.get_sliced_feature(feat, i + 5, [2, 2], 25)[1] for i in range(B)]
        out[f"chunk_idx_{n}"] = np.asarray(idxs)
        out[f"chunk_last_{n}"] = np.asarray(chunks[-1])
        assert all(c.shape == (50, 384) for c in chunks)
    np.savez_compressed(os.path.join(HERE, "whisper_golden.npz"), **out)
    print({k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
