"""`musetalk.models.vae` drop-in: `VAE.decode_latents` (musetalk/models/vae.py:96-108) on the hot path, and the encoder side of avatar
preparation (`preprocess_img` for in-memory crops, `encode_latents`, `get_latents_for_unet`, vae.py:52-94,110-122 -- called once per avatar
frame by mere_musetalk.py:303-304).

musereal.py:57-61,108 uses `vae.vae` (for `.half()`) and `vae.decode_latents(pred_latents)`, which must return a
uint8 ndarray (B, 256, 256, 3) in BGR.  Reading image FILES (cv2.imread + INTER_LANCZOS4 resize, vae.py:62-67) stays with the reference:
the avatar builder already hands over 256 x 256 crops (mere_musetalk.py:302).
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


def load_diffusers_weights(model_dir, stem="diffusion_pytorch_model"):
    """The weight file `from_pretrained(model_dir)` would read (vae.py:24): `<stem>.safetensors` when present, else `<stem>.bin`."""
    st = os.path.join(model_dir, stem + ".safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st, device="cpu")
    return torch.load(os.path.join(model_dir, stem + ".bin"), map_location="cpu")


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
            state_dict = load_diffusers_weights(model_path)
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
        self._precision = precision
        self._config = dict(config)
        # the encoder half is only needed for avatar preparation: its handle is created on first use
        self._enc_sd = {k: v for k, v in remap_legacy_attention_keys(state_dict).items() if k.startswith(("encoder.", "quant_conv."))}
        self._enc_h = None
        if use_float16:
            self.vae = self.vae.half()

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().mf_vae_destroy(self._h)
            if getattr(self, "_enc_h", None):
                _lib.lib().mf_vae_encoder_destroy(self._enc_h)
        except Exception:
            pass

    # ---- encoder side: avatar preparation (vae.py:40-94, 110-122) ---------------------------------------------------------------
    ENC_BATCH = 2

    def _encoder(self):
        if self._enc_h is None:
            if not self._enc_sd:
                raise RuntimeError("this VAE was built from a decoder-only state dict: no `encoder.*` / `quant_conv.*` tensors to encode with")
            arr, keep = _lib.tensor_array(self._enc_sd)
            h = C.c_void_p()
            _lib.check(_lib.lib().mf_vae_encoder_create(C.byref(self._cfg), arr, len(keep), _lib.PRECISIONS[self._precision], self.ENC_BATCH, C.byref(h)),
                       "vae_encoder_create")
            self._enc_h = h.value
        return self._enc_h

    def encode_moments_device(self, image=None, image_u8_bgr=None, half_mask=False):
        """(mean | logvar) of `vae.encode(image).latent_dist`: fp32 [B, 8, 32, 32] on the device.  `image`: normalised fp32 [B,3,256,256]
        (what vae.py:84 receives), or `image_u8_bgr`: uint8 [B,256,256,3] crops preprocessed on the device (vae.py:52-82)."""
        src = image if image is not None else image_u8_bgr
        if src is None or not src.is_cuda:
            raise RuntimeError("VAE.encode needs HIP device tensors; no CPU path exists here")
        h = self._encoder()
        B = src.shape[0]
        if B > self.ENC_BATCH:
            return torch.cat([self.encode_moments_device(None if image is None else image[i:i + self.ENC_BATCH],
                                                         None if image_u8_bgr is None else image_u8_bgr[i:i + self.ENC_BATCH], half_mask)
                              for i in range(0, B, self.ENC_BATCH)], dim=0)
        src = src.contiguous().float() if image is not None else src.contiguous()
        size = _lib.lib().mf_vae_encoder_image_size(h)
        want = (B, 3, size, size) if image is not None else (B, size, size, 3)
        if tuple(src.shape) != want or (image is None and src.dtype != torch.uint8):
            raise RuntimeError(f"VAE.encode: expected {'fp32' if image is not None else 'uint8'} {want}, got {src.dtype} {tuple(src.shape)}")
        mom = torch.empty((B, 2 * self._cfg.latent_channels, size // 8, size // 8), dtype=torch.float32, device=src.device)
        with torch.cuda.device(src.device):
            _lib.check(_lib.lib().mf_vae_encode(h, src.data_ptr() if image is not None else None, src.data_ptr() if image is None else None,
                                                int(bool(half_mask)), mom.data_ptr(), B, C.c_void_p(torch.cuda.current_stream(src.device).cuda_stream)),
                       "vae_encode")
        return mom

    def get_mask_tensor(self):
        """vae.py:40-50."""
        m = torch.zeros((self._resized_img, self._resized_img))
        m[:self._resized_img // 2, :] = 1
        return m

    def preprocess_img(self, img_name, half_mask=False):
        """vae.py:52-82 for an in-memory BGR crop (the only form mere_musetalk.py:303 passes): fp32 [1, 3, 256, 256] RGB on the device."""
        if isinstance(img_name, str):
            raise RuntimeError("VAE.preprocess_img: reading image files (cv2.imread + INTER_LANCZOS4) stays with the reference; pass the BGR crop")
        x = torch.from_numpy(np.ascontiguousarray(np.asarray(img_name)[:, :, ::-1] / 255.)).float().permute(2, 0, 1)
        if half_mask:
            x = x * (self.get_mask_tensor() > 0.5)
        x = (x - 0.5) / 0.5
        return x.unsqueeze(0).to(self.device)

    @staticmethod
    def _sample(moments, scaling_factor, generator=None):
        """diffusers DiagonalGaussianDistribution.sample() * scaling_factor (vae.py:92-93)."""
        mean, logvar = torch.chunk(moments, 2, dim=1)
        std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
        noise = torch.randn(mean.shape, generator=generator, device=mean.device, dtype=mean.dtype)
        return scaling_factor * (mean + std * noise)

    def encode_latents(self, image, generator=None):
        """vae.py:84-94: `scaling_factor * vae.encode(image).latent_dist.sample()`; the noise comes from torch's generator, as in diffusers."""
        return self._sample(self.encode_moments_device(image=image), self.scaling_factor, generator)

    def get_latents_for_unet(self, img, generator=None):
        """vae.py:110-122: [masked latents | reference latents] of one 256 x 256 BGR crop -> [1, 8, 32, 32] (what latents.pt holds).  Both
        encoder passes run as one batch of two; the two noise draws keep the reference's order (masked first)."""
        if isinstance(img, str):
            raise RuntimeError("VAE.get_latents_for_unet: pass the 256 x 256 BGR crop (mere_musetalk.py:302-303), not a file name")
        crop = torch.from_numpy(np.ascontiguousarray(img)).to(self.device)
        mom_masked = self.encode_moments_device(image_u8_bgr=crop[None], half_mask=True)
        mom_ref = self.encode_moments_device(image_u8_bgr=crop[None], half_mask=False)
        return torch.cat([self._sample(mom_masked, self.scaling_factor, generator), self._sample(mom_ref, self.scaling_factor, generator)], dim=1)

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

    def tune(self, batch):
        """Explicit launch-configuration warm-up of the decoder at this batch size (mf_vae_tune; see HipUNetModel.tune)."""
        _lib.check(_lib.lib().mf_vae_tune(self._h, int(batch), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "vae_tune")
