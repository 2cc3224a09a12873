"""ER-NeRF's audio front-end on the GPU: what `NerfASR` builds from transformers and calls every step (nerfasr.py:38-45,128-143).

    self.processor = AutoProcessor.from_pretrained(opt.asr_model)            -> RawProcessor        (the normalisation moves into the device call)
    self.model = AutoModelForCTC.from_pretrained(opt.asr_model).to(device)   -> HipWav2Vec2ForCTC   (`model(input_values).logits`)
    HubertModel.from_pretrained(...)                                         -> HipWav2Vec2ForCTC(..., out_hidden=True) (`.last_hidden_state`)

Same call shapes and return attributes, so `__frame_to_text` runs unchanged on them; `NerfASRFrontend` is the queue-free restatement of
run_step / get_next_feat (nerfasr.py:75-126) with the feature ring resident on the device.  The network runs in `mf_wav2vec2_forward`
(csrc/mf_wav2vec2.hip); there is no CPU path."""
import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch

from .. import _lib


def _cfg_get(cfg, k):
    return cfg[k] if isinstance(cfg, dict) else getattr(cfg, k)


class RawProcessor:
    """Stands where `Wav2Vec2Processor` / `AutoProcessor` stands in nerfasr.py:131: same call, `.input_values` [1, n] -- the raw samples;
    zero-mean / unit-variance normalisation (Wav2Vec2FeatureExtractor, do_normalize) happens inside mf_wav2vec2_forward."""

    def __call__(self, frame, sampling_rate=16000, return_tensors="pt", padding=True):
        if sampling_rate != 16000:
            raise ValueError("the wav2vec2 front-end is built for 16 kHz audio (basereal.py:38)")
        x = torch.as_tensor(np.asarray(frame, dtype=np.float32))
        return SimpleNamespace(input_values=x.reshape(1, -1) if x.dim() == 1 else x)


class HipWav2Vec2ForCTC:
    def __init__(self, config, state_dict, max_windows=1, precision="bf16x3", device="cuda", out_hidden=False, do_normalize=True):
        """config: transformers' Wav2Vec2Config / HubertConfig or a dict with the same field names; state_dict: the model's own state dict."""
        if not torch.cuda.is_available():
            raise RuntimeError("HipWav2Vec2ForCTC needs a HIP device; no CPU path exists here")
        if _cfg_get(config, "feat_extract_norm") != "layer":
            raise ValueError('only feat_extract_norm == "layer" checkpoints are supported (the xlsr-53 / large models app.py:660-662 names)')
        self.device = torch.device(device)
        _lib.init_device(self.device.index if self.device.index is not None else torch.cuda.current_device())
        c = _lib.MfWav2Vec2Config()
        c.hidden, c.n_layer, c.n_head = _cfg_get(config, "hidden_size"), _cfg_get(config, "num_hidden_layers"), _cfg_get(config, "num_attention_heads")
        c.ffn, c.vocab = _cfg_get(config, "intermediate_size"), 0 if out_hidden else _cfg_get(config, "vocab_size")
        dims, ks, ss = list(_cfg_get(config, "conv_dim")), list(_cfg_get(config, "conv_kernel")), list(_cfg_get(config, "conv_stride"))
        c.n_conv = len(dims)
        for i in range(c.n_conv):
            c.conv_dim[i], c.conv_kernel[i], c.conv_stride[i] = dims[i], ks[i], ss[i]
        c.conv_bias, c.feat_norm_layer = int(bool(_cfg_get(config, "conv_bias"))), 1
        c.stable_ln = int(bool(_cfg_get(config, "do_stable_layer_norm")))
        c.pos_k, c.pos_groups = _cfg_get(config, "num_conv_pos_embeddings"), _cfg_get(config, "num_conv_pos_embedding_groups")
        c.layer_norm_eps = float(_cfg_get(config, "layer_norm_eps"))
        c.do_normalize, c.out_hidden = int(do_normalize), int(out_hidden)
        self._cfg, self._sd = c, {k: v for k, v in state_dict.items() if torch.is_tensor(v) and v.is_floating_point()}
        self._precision, self._max_windows, self.out_hidden = precision, max_windows, out_hidden
        self._h, self._n = None, 0
        self.config = config

    @classmethod
    def from_hf(cls, model, **kw):
        """From an instantiated transformers model (Wav2Vec2ForCTC / HubertModel) -- what `from_pretrained` returns in nerfasr.py:42-45."""
        return cls(model.config, model.state_dict(), out_hidden=not hasattr(model, "lm_head"), **kw)

    def to(self, device):            # nerfasr.py:45 `.to(self.device)`
        return self

    def eval(self):
        return self

    def _handle(self, n):
        if self._h is not None and self._n == n:
            return self._h
        if self._h is not None:
            _lib.lib().mf_wav2vec2_destroy(self._h)
            self._h = None
        arr, keep = _lib.tensor_array(self._sd)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mf_wav2vec2_create(C.byref(self._cfg), arr, len(keep), int(n), int(self._max_windows),
                                                     _lib.PRECISIONS[self._precision], C.byref(h)), "wav2vec2_create")
        self._h, self._n = h.value, n
        t, w = C.c_int(), C.c_int()
        _lib.check(_lib.lib().mf_wav2vec2_frames(self._h, C.byref(t), C.byref(w)))
        self.n_frames, self.width = t.value, w.value
        return self._h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().mf_wav2vec2_destroy(self._h)
        except Exception:
            pass

    def __call__(self, input_values):
        """input_values: [S, n] float32 (a CPU or device tensor) -> namespace with `.logits` [S, T, vocab] (or `.last_hidden_state`) on the device"""
        x = torch.as_tensor(input_values).to(self.device, torch.float32)
        if x.dim() == 1:
            x = x[None]
        x = x.contiguous()
        S, n = x.shape
        h = self._handle(n)
        out = torch.empty((S, self.n_frames, self.width), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mf_wav2vec2_forward(h, x.data_ptr(), n, S, out.data_ptr(),
                                                      C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "wav2vec2_forward")
        return SimpleNamespace(last_hidden_state=out) if self.out_hidden else SimpleNamespace(logits=out)


class NerfASRFrontend:
    """nerfasr.py:15-126 without the queues: `put_audio_frame` 20 ms chunks, `run_step()` as the reference's, `get_next_feat()` -> [8, dim, 16]
    (att > 0) or [1, dim, 16].  The feature ring `feat_queue` (feat_buffer_size * m rows) stays on the device."""

    def __init__(self, model, processor=None, m=8, l=10, r=10, fps=50, att=2, audio_dim=44, device="cuda"):
        self.model, self.processor = model, processor or RawProcessor()
        self.context_size, self.stride_left_size, self.stride_right_size = m, l, r          # nerfasr.py:30-32
        self.chunk = 16000 // fps
        self.att, self.audio_dim, self.device = att, audio_dim, torch.device(device)
        self.frames = [np.zeros(self.chunk, np.float32)] * l if l > 0 else []                # :35-36
        self.feat_buffer_size, self.feat_buffer_idx = 4, 0                                    # :48-49
        self.feat_queue = torch.zeros(self.feat_buffer_size * m, audio_dim, dtype=torch.float32, device=self.device)
        self.front, self.tail = self.feat_buffer_size * m - 8, 8                              # :52-53
        self.att_feats = [torch.zeros(audio_dim, 16, dtype=torch.float32, device=self.device)] * 4   # :55
        self.warm_up_steps = m + l + r                                                        # :58
        self.pending = []

    def put_audio_frame(self, frame):
        self.pending.append(np.asarray(frame, np.float32))

    def _next_frame(self):
        return self.pending.pop(0) if self.pending else np.zeros(self.chunk, np.float32)      # :67-71 (silence when the queue is empty)

    def frame_to_logits(self, frame):
        """nerfasr.py:128-143"""
        inputs = self.processor(frame, sampling_rate=16000, return_tensors="pt", padding=True)
        result = self.model(inputs.input_values.to(self.device))
        logits = result.last_hidden_state if hasattr(result, "last_hidden_state") else result.logits
        left = max(0, self.stride_left_size)
        right = min(logits.shape[1], logits.shape[1] - self.stride_right_size + 1)
        return logits[:, left:right][0]

    def run_step(self):
        """nerfasr.py:105-124"""
        self.frames.append(self._next_frame())
        if len(self.frames) < self.stride_left_size + self.context_size + self.stride_right_size:
            return
        inputs = np.concatenate(self.frames)
        self.frames = self.frames[-(self.stride_left_size + self.stride_right_size):]
        feats = self.frame_to_logits(inputs)
        start = self.feat_buffer_idx * self.context_size
        self.feat_queue[start:start + feats.shape[0]] = feats
        self.feat_buffer_idx = (self.feat_buffer_idx + 1) % self.feat_buffer_size

    def _window(self):
        if self.front < self.tail:
            feat = self.feat_queue[self.front:self.tail]
        else:
            feat = torch.cat([self.feat_queue[self.front:], self.feat_queue[:self.tail]], dim=0)
        self.front = (self.front + 2) % self.feat_queue.shape[0]
        self.tail = (self.tail + 2) % self.feat_queue.shape[0]
        return feat.permute(1, 0)

    def get_next_feat(self):
        """nerfasr.py:75-103"""
        if self.att > 0:
            while len(self.att_feats) < 8:
                self.att_feats.append(self._window())
            att_feat = torch.stack(self.att_feats, dim=0)
            self.att_feats = self.att_feats[1:]
            return att_feat
        return self._window().unsqueeze(0)
