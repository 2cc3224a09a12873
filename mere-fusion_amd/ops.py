"""PyTorch-ROCm custom ops over the C ABI (include/merefusion.h).

torch is plumbing here: it owns device memory and the current HIP stream; every op hands raw
device pointers to libmerefusion_hip.so.  Ops refuse CPU tensors -- there is no fallback path.
"""
import ctypes as C

import torch

from . import _lib


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_cuda(name, *tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(
                f"merefusion::{name}: tensor is on {t.device}; the MI355X path needs HIP device tensors "
                "(no CPU fallback is provided)")


@torch.library.custom_op("merefusion::wav2lip_forward", mutates_args=())
def wav2lip_forward(handle: int, mel: torch.Tensor, face: torch.Tensor) -> torch.Tensor:
    """pred = model(mel_batch, img_batch)  (lipreal.py:124-125).  mel [B,1,80,16], face [B,6,96,96]."""
    _require_cuda("wav2lip_forward", mel, face)
    if mel.dim() != 4 or tuple(mel.shape[1:]) != (1, 80, 16):
        raise RuntimeError(f"wav2lip_forward: mel must be [B,1,80,16], got {tuple(mel.shape)}")
    if face.dim() != 4 or tuple(face.shape[1:]) != (6, 96, 96) or face.shape[0] != mel.shape[0]:
        raise RuntimeError(f"wav2lip_forward: face must be [B,6,96,96] with B={mel.shape[0]}, got {tuple(face.shape)}")
    mel = mel.contiguous().float()
    face = face.contiguous().float()
    B = mel.shape[0]
    out = torch.empty((B, 3, 96, 96), dtype=torch.float32, device=mel.device)
    if B == 0:
        return out
    with torch.cuda.device(mel.device):
        _lib.check(_lib.lib().mf_wav2lip_forward(handle, mel.data_ptr(), face.data_ptr(), out.data_ptr(), B,
                                                 _stream_ptr(mel.device)), "wav2lip_forward")
    return out


@wav2lip_forward.register_fake
def _(handle, mel, face):
    return mel.new_empty((mel.shape[0], 3, 96, 96), dtype=torch.float32)


@torch.library.custom_op("merefusion::wav2lip_forward_u8", mutates_args=())
def wav2lip_forward_u8(handle: int, mel: torch.Tensor, faces_u8: torch.Tensor) -> torch.Tensor:
    """lipreal.py:115-126 fused: uint8 [B,96,96,3] BGR crops + mel [B,1,80,16] -> fp32 [B,96,96,3] = pred*255."""
    _require_cuda("wav2lip_forward_u8", mel, faces_u8)
    if faces_u8.dtype != torch.uint8 or faces_u8.dim() != 4 or tuple(faces_u8.shape[1:]) != (96, 96, 3):
        raise RuntimeError(f"wav2lip_forward_u8: faces must be uint8 [B,96,96,3], got {faces_u8.dtype} {tuple(faces_u8.shape)}")
    if mel.dim() != 4 or tuple(mel.shape[1:]) != (1, 80, 16) or mel.shape[0] != faces_u8.shape[0]:
        raise RuntimeError(f"wav2lip_forward_u8: mel must be [B,1,80,16], got {tuple(mel.shape)}")
    mel = mel.contiguous().float()
    faces_u8 = faces_u8.contiguous()
    B = mel.shape[0]
    out = torch.empty((B, 96, 96, 3), dtype=torch.float32, device=mel.device)
    if B == 0:
        return out
    with torch.cuda.device(mel.device):
        _lib.check(_lib.lib().mf_wav2lip_forward_u8(handle, mel.data_ptr(), faces_u8.data_ptr(), out.data_ptr(), B,
                                                    _stream_ptr(mel.device)), "wav2lip_forward_u8")
    return out


@wav2lip_forward_u8.register_fake
def _(handle, mel, faces_u8):
    return mel.new_empty((mel.shape[0], 96, 96, 3), dtype=torch.float32)


@torch.library.custom_op("merefusion::melspec", mutates_args=())
def melspec(wav: torch.Tensor, pad_mode: int) -> torch.Tensor:
    """audio.melspectrogram (wav2lip/audio.py:45-51): fp32 [n] -> fp32 [80, 1 + n//200]."""
    _require_cuda("melspec", wav)
    if wav.dim() != 1:
        raise RuntimeError(f"melspec: wav must be 1-D, got {tuple(wav.shape)}")
    wav = wav.contiguous().float()
    n = wav.shape[0]
    T = _lib.lib().mf_melspec_frames(n)
    out = torch.empty((80, T), dtype=torch.float32, device=wav.device)
    with torch.cuda.device(wav.device):
        _lib.check(_lib.lib().mf_melspec(wav.data_ptr(), n, out.data_ptr(), int(pad_mode), _stream_ptr(wav.device)),
                   "melspec")
    return out


@melspec.register_fake
def _(wav, pad_mode):
    return wav.new_empty((80, 1 + wav.shape[0] // 200), dtype=torch.float32)
