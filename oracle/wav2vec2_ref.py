"""TEST INFRASTRUCTURE (oracle/): the CPU side of NerfASR.__frame_to_text (nerfasr.py:128-143) for the GPU parity tests of
mere-fusion_amd/csrc/mf_wav2vec2.hip.  Never imported by the product.

The arithmetic of this stage lives in a third-party dependency of the reference -- transformers (requirements.txt, unpinned; 5.15.0 in this
image, on the GPU box too) -- which nerfasr.py:41-45 instantiates as AutoProcessor + AutoModelForCTC.  The dependency is PRESENT, so the oracle
is the dependency itself: `Wav2Vec2FeatureExtractor(do_normalize=True)` and `Wav2Vec2ForCTC(Wav2Vec2Config(...))` on the CPU in fp32, loaded
with the seeded state dict under `strict=True` (which also pins the key names / shapes the C loader reads against the library).  PINNED.
No checkpoint exists here (no network): architecture = the published XLSR-53 large config with the 44-symbol head nerfasr.py:19-20 expects."""
import numpy as np
import torch


def build(cfg, state_dict):
    from transformers import Wav2Vec2Config, Wav2Vec2ForCTC
    hf = Wav2Vec2Config(**{k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()},
                        hidden_dropout=0.0, activation_dropout=0.0, attention_dropout=0.0, feat_proj_dropout=0.0, final_dropout=0.0, layerdrop=0.0,
                        pad_token_id=0)
    model = Wav2Vec2ForCTC(hf)
    missing, unexpected = model.load_state_dict(state_dict, strict=True)
    return model.eval()


def frame_to_logits(model, wav):
    """wav: float32 [n] or [S, n] raw samples -> logits [S, T, vocab] (processor + model(...).logits, nerfasr.py:131-136)"""
    from transformers import Wav2Vec2FeatureExtractor
    fe = Wav2Vec2FeatureExtractor(feature_size=1, sampling_rate=16000, padding_value=0.0, do_normalize=True, return_attention_mask=False)
    wav = np.atleast_2d(np.asarray(wav, np.float32))
    inputs = fe([w for w in wav], sampling_rate=16000, return_tensors="pt", padding=True)
    with torch.no_grad():
        return model(inputs.input_values).logits.numpy()


def slice_logits(logits, stride_left, stride_right):
    """nerfasr.py:138-141"""
    left = max(0, stride_left)
    right = min(logits.shape[1], logits.shape[1] - stride_right + 1)
    return logits[:, left:right]
