"""`NeRFNetwork.encode_audio` (ernerf/nerf_triplane/network.py:222-237) on MI355X: AudioNet + AudioAttNet in one kernel launch.

    enc = HipAudioEncoder(model.state_dict(), att=opt.att)
    model.encode_audio = enc.encode_audio          # renderer.py:187 calls self.encode_audio(auds)
"""
import ctypes as C

import torch

from .. import _lib


class HipAudioEncoder:
    def __init__(self, state_dict, att=2, device="cuda"):
        self.device = torch.device(device)
        _lib.init_device(self.device.index or 0)
        self._lib = _lib.lib()
        keep = {k: v for k, v in state_dict.items() if k.startswith(("audio_net.", "audio_att_net."))}
        arr, self._keep = _lib.tensor_array(keep)
        self._h = C.c_void_p()
        self.att = int(att)
        _lib.check(self._lib.mf_audio_encoder_create(arr, len(arr), int(self.att > 0), C.byref(self._h)), "mf_audio_encoder_create")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.mf_audio_encoder_destroy(h)
            self._h = None

    def encode_audio_smooth(self, a, prev):
        """`encode_audio(a)` followed by renderer.py:190-194's lip smoothing against the previous frame's features `prev` ([1, 32] CUDA fp32, or None on
        the first frame): 0.35 * prev + (1 - 0.35) * enc_a, inside the same launch (bit-identical to the torch expression)."""
        if a is None:
            return None
        if not (torch.is_tensor(a) and a.is_cuda):
            raise RuntimeError("HipAudioEncoder.encode_audio_smooth: the window must be a CUDA tensor (there is no CPU path)")
        a = a.float().contiguous()
        out = torch.empty(1, 32, device=a.device)
        pv = None
        if prev is not None:
            pv = prev.float().contiguous()
            if pv.numel() != 32 or not pv.is_cuda:
                raise RuntimeError("HipAudioEncoder.encode_audio_smooth: prev must be the previous [1, 32] CUDA feature vector")
        _lib.check(self._lib.mf_audio_encoder_forward_smooth(self._h, C.c_void_p(a.data_ptr()), int(a.shape[0]),
                                                             C.c_void_p(pv.data_ptr()) if pv is not None else None, C.c_void_p(out.data_ptr()),
                                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), "mf_audio_encoder_forward_smooth")
        return out

    def encode_audio(self, a):
        """a: [8, audio_in_dim, 16] (att > 0) or [1, audio_in_dim, 16] CUDA tensor -> [1, 32]; None passes through (network.py:227)."""
        if a is None:
            return None
        if not (torch.is_tensor(a) and a.is_cuda):
            raise RuntimeError("HipAudioEncoder.encode_audio: the window must be a CUDA tensor (there is no CPU path)")
        a = a.float().contiguous()
        out = torch.empty(1, 32, device=a.device)
        _lib.check(self._lib.mf_audio_encoder_forward(self._h, C.c_void_p(a.data_ptr()), int(a.shape[0]), C.c_void_p(out.data_ptr()),
                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)), "mf_audio_encoder_forward")
        return out
