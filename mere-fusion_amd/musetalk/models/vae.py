"""`musetalk.models.vae` drop-in, hot-path subset: `VAE.decode_latents` (musetalk/models/vae.py:96-108).

musereal.py:57-61,108 uses `vae.vae` (for `.half()`) and `vae.decode_latents(pred_latents)`, which must return a
uint8 ndarray (B, 256, 256, 3) in BGR.  Encoding (avatar preparation, vae.py:84-94,110-122) is offline and
stays with the reference.
"""
import ctypes as C
import json
import os

import numpy as np
import torch

from ... import _lib, ops

SD_VAE_FT_MSE = dict(latent_channels=4, out_channels=3, block_out_channels=[128, 256, 512, 512], layers_per_block=2,
                     norm_num_groups=32, scaling_factor=0.18215, sample_size=32)


# diffusers renamed the VAE attention block's parameters (AttentionBlock -> Attention); `AutoencoderKL.from_pretrained` -- what the reference
# calls (vae.py:24) -- converts old checkpoints on load, and the published sd-vae-ft-mse file carries the OLD names.
_LEGACY_ATTN_KEYS = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def remap_legacy_attention_keys(state_dict):
    """`...attentions.N.{query,key,value,proj_attn}.{weight,bias}` -> `...attentions.N.{to_q,to_k,to_v,to_out.0}.{weight,bias}`;
    new-style keys pass through untouched.  Old checkpoints may store the projections as 1x1 convolutions [C, C, 1, 1]: same element
    order as the Linear [C, C] the C loader expects."""
    out = {}
    for k, v in state_dict.items():
        parts = k.split(".")
        if len(parts) >= 3 and "attentions" in parts and parts[-2] in _LEGACY_ATTN_KEYS:
            k = ".".join(parts[:-2] + [_LEGACY_ATTN_KEYS[parts[-2]], parts[-1]])
        out[k] = v
    return out


def vae_config_struct(cfg):
    boc = list(cfg["block_out_channels"])
    c = _lib.MfVaeConfig()
    c.latent_channels, c.out_channels, c.n_blocks = cfg.get("latent_channels", 4), cfg.get("out_channels", 3), len(boc)
    for i, v in enumerate(boc):
        c.block_out_channels[i] = v
    c.layers_per_block, c.norm_num_groups = cfg.get("layers_per_block", 2), cfg.get("norm_num_groups", 32)
    # a diffusers AutoencoderKL config stores the IMAGE size in sample_size; the latent grid is what matters here
    c.sample_size = cfg.get("latent_size", 32)
    c.scaling_factor = cfg.get("scaling_factor", 0.18215)
    return c


class _HipVaeModule:
    """Stands where `AutoencoderKL` stood for the attributes musereal.py touches."""

    def __init__(self):
        self.dtype = torch.float32

    def half(self):
        self.dtype = torch.float16
        return self

    def to(self, *a, **k):
        return self


class VAE:
    def __init__(self, model_path="./models/sd-vae-ft-mse/", resized_img=256, use_float16=False, config=None,
                 state_dict=None, precision="bf16x3", max_batch=16):
        if state_dict is None:
            with open(os.path.join(model_path, "config.json")) as f:
                config = json.load(f)
            wpath = os.path.join(model_path, "diffusion_pytorch_model.bin")
            state_dict = torch.load(wpath, map_location="cpu")
        if not torch.cuda.is_available():
            raise RuntimeError("the MuseTalk VAE decoder needs a HIP device; no CPU path exists here")
        self.model_path = model_path
        self.device = torch.device("cuda")
        self.vae = _HipVaeModule()
        self.scaling_factor = config.get("scaling_factor", 0.18215)
        self._resized_img = resized_img
        _lib.init_device(torch.cuda.current_device())
        self._cfg = vae_config_struct(config)
        arr, keep = _lib.tensor_array(remap_legacy_attention_keys(state_dict))
        h = C.c_void_p()
        _lib.check(_lib.lib().mf_vae_create(C.byref(self._cfg), arr, len(keep), _lib.PRECISIONS[precision], int(max_batch), C.byref(h)),
                   "vae_create")
        self._h = h.value
        self.max_batch = max_batch
        if use_float16:
            self.vae = self.vae.half()

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().mf_vae_destroy(self._h)
        except Exception:
            pass

    def decode_latents_device(self, latents, want_image=False):
        """uint8 frames [B, 8S, 8S, 3] BGR on the device (+ the pre-clamp fp32 image [B,3,8S,8S] if asked)."""
        if not latents.is_cuda:
            raise RuntimeError("VAE.decode_latents needs HIP device tensors; no CPU path exists here")
        if not want_image:
            return ops.vae_decode_latents(self._h, latents)               # merefusion::vae_decode_latents
        lat = latents.float().contiguous()
        B, S = lat.shape[0], lat.shape[2] * 8
        frames = torch.empty((B, S, S, 3), dtype=torch.uint8, device=lat.device)
        image = torch.empty((B, 3, S, S), dtype=torch.float32, device=lat.device) if want_image else None
        with torch.cuda.device(lat.device):
            _lib.check(_lib.lib().mf_vae_decode_latents(self._h, lat.data_ptr(), frames.data_ptr(),
                                                        image.data_ptr() if want_image else None, B,
                                                        C.c_void_p(torch.cuda.current_stream(lat.device).cuda_stream)),
                       "vae_decode_latents")
        return (frames, image) if want_image else frames

    def decode_latents(self, latents):
        return self.decode_latents_device(latents).cpu().numpy()
