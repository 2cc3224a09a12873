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


# ---- MuseTalk / Whisper stages as custom ops too (the drop-in modules call these; handles are the C ABI's opaque pointers) ----------
@torch.library.custom_op("merefusion::unet_forward", mutates_args=())
def unet_forward(handle: int, latents: torch.Tensor, audio: torch.Tensor, add_pe: bool, out_channels: int) -> torch.Tensor:
    """unet.model(latent_batch, timesteps=[0], encoder_hidden_states=audio).sample (musereal.py:105-107): latents [B,8,S,S], audio [B,T,384]
    (+ the positional encoding of unet.py:12-27 when add_pe) -> [B,4,S,S] fp32."""
    _require_cuda("unet_forward", latents, audio)
    if latents.dim() != 4 or audio.dim() != 3 or audio.shape[0] != latents.shape[0]:
        raise RuntimeError(f"unet_forward: latents [B,C,S,S] and audio [B,T,D] expected, got {tuple(latents.shape)} / {tuple(audio.shape)}")
    lat, aud = latents.contiguous().float(), audio.contiguous().float()
    B = lat.shape[0]
    out = torch.empty((B, out_channels, lat.shape[2], lat.shape[3]), dtype=torch.float32, device=lat.device)
    if B == 0:
        return out
    with torch.cuda.device(lat.device):
        _lib.check(_lib.lib().mf_unet_forward(handle, lat.data_ptr(), aud.data_ptr(), int(add_pe), out.data_ptr(), B, _stream_ptr(lat.device)),
                   "unet_forward")
    return out


@unet_forward.register_fake
def _(handle, latents, audio, add_pe, out_channels):
    return latents.new_empty((latents.shape[0], out_channels, latents.shape[2], latents.shape[3]), dtype=torch.float32)


@torch.library.custom_op("merefusion::vae_decode_latents", mutates_args=())
def vae_decode_latents(handle: int, latents: torch.Tensor) -> torch.Tensor:
    """VAE.decode_latents (musetalk/models/vae.py:96-108) up to the host copy: latents [B,4,S,S] -> uint8 BGR frames [B,8S,8S,3] on the device."""
    _require_cuda("vae_decode_latents", latents)
    if latents.dim() != 4:
        raise RuntimeError(f"vae_decode_latents: latents [B,4,S,S] expected, got {tuple(latents.shape)}")
    lat = latents.contiguous().float()
    B, S = lat.shape[0], lat.shape[2] * 8
    frames = torch.empty((B, S, S, 3), dtype=torch.uint8, device=lat.device)
    if B == 0:
        return frames
    with torch.cuda.device(lat.device):
        _lib.check(_lib.lib().mf_vae_decode_latents(handle, lat.data_ptr(), frames.data_ptr(), None, B, _stream_ptr(lat.device)), "vae_decode_latents")
    return frames


@vae_decode_latents.register_fake
def _(handle, latents):
    return latents.new_empty((latents.shape[0], latents.shape[2] * 8, latents.shape[3] * 8, 3), dtype=torch.uint8)


@torch.library.custom_op("merefusion::whisper_encode_windows", mutates_args=())
def whisper_encode_windows(handle: int, wavs: torch.Tensor, ctx_tokens: int, n_layers1: int, n_state: int) -> torch.Tensor:
    """Audio2Feature.audio2feat for S windows in one encoder call (museasr.py:25-26 -> audio2feature.py:99-112): wavs fp32 [S, n] ->
    [S, n // 320, n_layer + 1, n_state].  The handle's workspace must hold S windows (mf_whisper_set_batch)."""
    _require_cuda("whisper_encode_windows", wavs)
    if wavs.dim() != 2:
        raise RuntimeError(f"whisper_encode_windows: wavs [S, n] expected, got {tuple(wavs.shape)}")
    w = wavs.contiguous().float()
    S, n = w.shape
    feat = torch.empty((S, n // 320, n_layers1, n_state), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(_lib.lib().mf_whisper_encode_windows(handle, w.data_ptr(), n, S, int(ctx_tokens), feat.data_ptr(), _stream_ptr(w.device)),
                   "whisper_encode_windows")
    return feat


@whisper_encode_windows.register_fake
def _(handle, wavs, ctx_tokens, n_layers1, n_state):
    return wavs.new_empty((wavs.shape[0], wavs.shape[1] // 320, n_layers1, n_state), dtype=torch.float32)
