"""`musetalk.models.unet` drop-in: `UNet` and `PositionalEncoding` (musetalk/models/unet.py:12-44).

musereal.py:57-62,100-107 uses: `unet.model`, `unet.device`, `unet.model.dtype`, `unet.model.half()`,
`pe.half()`, `pe(audio)`, `unet.model(latents, timesteps, encoder_hidden_states=audio).sample`.
The HIP UNet fuses `pe(audio)` and fixes the timestep to 0 (musereal.py:59); both objects keep their call shape.
"""
import ctypes as C
import json

import torch

from ... import _lib, ops

# diffusers config of the public MuseTalk v1 UNet [upstream-knowledge, SURVEY Appendix C]; the reference reads
# ./models/musetalk/musetalk.json (musetalk/utils/utils.py:69), which does not ship
MUSETALK_V1_UNET = dict(in_channels=8, out_channels=4, block_out_channels=[320, 640, 1280, 1280], layers_per_block=2,
                        cross_attention_dim=384, attention_head_dim=8, norm_num_groups=32,
                        down_block_types=["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"],
                        up_block_types=["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3, sample_size=32)


def unet_config_struct(cfg, ctx_len=50):
    boc = list(cfg["block_out_channels"])
    c = _lib.MfUnetConfig()
    c.in_channels, c.out_channels, c.n_blocks = cfg["in_channels"], cfg["out_channels"], len(boc)
    c.layers_per_block, c.cross_attention_dim = cfg.get("layers_per_block", 2), cfg["cross_attention_dim"]
    heads = cfg.get("attention_head_dim", cfg.get("attention_heads", 8))
    c.attention_heads = heads if isinstance(heads, int) else heads[0]
    c.norm_num_groups = cfg.get("norm_num_groups", 32)
    down = cfg.get("down_block_types") or ["CrossAttnDownBlock2D" if a else "DownBlock2D" for a in cfg["down_attn"]]
    up = cfg.get("up_block_types") or ["CrossAttnUpBlock2D" if a else "UpBlock2D" for a in cfg["up_attn"]]
    for i in range(len(boc)):
        c.block_out_channels[i] = boc[i]
        c.down_attn[i] = int("CrossAttn" in down[i])
        c.up_attn[i] = int("CrossAttn" in up[i])
    c.sample_size, c.ctx_len = cfg.get("sample_size", 32), ctx_len
    return c


class PositionalEncoding(torch.nn.Module):
    """unet.py:12-27.  Returns its input unchanged when the UNet model it is paired with adds the encoding on the
    GPU (`fused=True`, what load_diffusion_model builds); stand-alone it adds the table itself."""

    def __init__(self, d_model=384, max_len=5000, fused=False):
        super().__init__()
        import math
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0))
        self.fused = fused

    def forward(self, x):
        if self.fused:
            return x
        return x + self.pe[:, :x.size(1), :].to(x.device)


class _Sample:
    def __init__(self, sample):
        self.sample = sample


class HipUNetModel:
    """Stands where `UNet2DConditionModel` stood: callable, `.dtype`, `.half()`, `.to()`."""

    def __init__(self, config, state_dict, precision="bf16x3", max_batch=16, device="cuda", fuse_pe=True):
        if not torch.cuda.is_available():
            raise RuntimeError("the MuseTalk UNet needs a HIP device; no CPU path exists here")
        self.device = torch.device(device)
        self.config = dict(config)
        self.dtype = torch.float32
        self.fuse_pe = fuse_pe
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.init_device(idx)
        self._cfg = unet_config_struct(config)
        arr, keep = _lib.tensor_array(state_dict)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mf_unet_create(C.byref(self._cfg), arr, len(keep), _lib.PRECISIONS[precision], int(max_batch), C.byref(h)),
                       "unet_create")
        self._h = h.value
        self.max_batch = max_batch

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().mf_unet_destroy(self._h)
        except Exception:
            pass

    def tune(self, batch):
        """Explicit launch-configuration warm-up (mf_unet_tune): times every implicit-GEMM layer at this batch size on the buffers of the last
        forward at that size and keeps the fastest; a server calls it at start-up for every batch size its loop emits, or ships MF_TUNE_CACHE.
        A forward itself never measures."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mf_unet_tune(self._h, int(batch), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "unet_tune")

    def half(self):   # musereal.py:62; arithmetic mode is fixed at create time, dtype only labels the I/O tensors
        self.dtype = torch.float16
        return self

    def to(self, *a, **k):
        return self

    def __call__(self, sample, timestep, encoder_hidden_states=None, **kw):
        t = torch.as_tensor(timestep).reshape(-1)
        if (t != 0).any():
            raise RuntimeError("the MI355X MuseTalk UNet is built for timesteps=[0] (musereal.py:59)")
        if not sample.is_cuda or not encoder_hidden_states.is_cuda:
            raise RuntimeError("UNet needs HIP device tensors; no CPU path exists here")
        if sample.shape[0] > self.max_batch:
            raise RuntimeError(f"unet_forward: batch {sample.shape[0]} exceeds the handle's max_batch {self.max_batch}")
        out = ops.unet_forward(self._h, sample, encoder_hidden_states, bool(self.fuse_pe), int(self._cfg.out_channels))   # merefusion::unet_forward
        return _Sample(out.to(self.dtype))


class UNet:
    def __init__(self, unet_config, model_path, use_float16=False, precision="bf16x3", max_batch=16):
        if isinstance(unet_config, str):
            with open(unet_config, "r") as f:
                unet_config = json.load(f)
        weights = model_path if isinstance(model_path, dict) else torch.load(model_path, map_location="cpu")
        self.device = torch.device("cuda")
        self.model = HipUNetModel(unet_config, weights, precision=precision, max_batch=max_batch, device=self.device)
        self.pe = PositionalEncoding(d_model=unet_config["cross_attention_dim"], fused=True)
        if use_float16:
            self.model = self.model.half()
