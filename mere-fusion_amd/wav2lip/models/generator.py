"""`Wav2Lip` drop-in: the reference's constructor / load_state_dict / forward surface
(wav2lip/models/wav2lip.py:8-125) over the MI355X HIP generator.

The module tree below exists only to carry parameters under the reference's state-dict keys
(`face_encoder_blocks.{b}.{i}.conv_block.{0|1}.*`, ..., `output_block.1.{weight,bias}`), so
`model.load_state_dict(new_s); model.to(device).eval()` (lipreal.py:43-53) works unchanged.
No layer here has a torch forward: `forward()` hands device pointers to libmerefusion_hip.so
through the `merefusion::wav2lip_forward` custom op and raises if that is impossible.
"""
import ctypes as C
import os

import torch
from torch import nn

from ... import _lib, ops, placement


class _Block(nn.Module):
    """Parameter holder with the key layout of conv.py:5-19 / 33-44 (conv_block.0, conv_block.1)."""

    def __init__(self, cin, cout, k, transposed=False):
        super().__init__()
        conv = nn.ConvTranspose2d(cin, cout, k) if transposed else nn.Conv2d(cin, cout, k)
        self.conv_block = nn.Sequential(conv, nn.BatchNorm2d(cout))

    def forward(self, *a, **kw):  # pragma: no cover
        raise RuntimeError("Wav2Lip layers have no torch forward; call the Wav2Lip module itself")


def _seq(*specs):
    return nn.Sequential(*[_Block(ci, co, k, t) for (ci, co, k, t) in specs])


class Wav2Lip(nn.Module):
    def __init__(self, precision=None):
        super().__init__()
        c, t = False, True
        self.face_encoder_blocks = nn.ModuleList([
            _seq((6, 16, 7, c)),
            _seq((16, 32, 3, c), (32, 32, 3, c), (32, 32, 3, c)),
            _seq((32, 64, 3, c), (64, 64, 3, c), (64, 64, 3, c), (64, 64, 3, c)),
            _seq((64, 128, 3, c), (128, 128, 3, c), (128, 128, 3, c)),
            _seq((128, 256, 3, c), (256, 256, 3, c), (256, 256, 3, c)),
            _seq((256, 512, 3, c), (512, 512, 3, c)),
            _seq((512, 512, 3, c), (512, 512, 1, c)),
        ])
        self.audio_encoder = _seq(
            (1, 32, 3, c), (32, 32, 3, c), (32, 32, 3, c), (32, 64, 3, c), (64, 64, 3, c), (64, 64, 3, c),
            (64, 128, 3, c), (128, 128, 3, c), (128, 128, 3, c), (128, 256, 3, c), (256, 256, 3, c),
            (256, 512, 3, c), (512, 512, 1, c))
        self.face_decoder_blocks = nn.ModuleList([
            _seq((512, 512, 1, c)),
            _seq((1024, 512, 3, t), (512, 512, 3, c)),
            _seq((1024, 512, 3, t), (512, 512, 3, c), (512, 512, 3, c)),
            _seq((768, 384, 3, t), (384, 384, 3, c), (384, 384, 3, c)),
            _seq((512, 256, 3, t), (256, 256, 3, c), (256, 256, 3, c)),
            _seq((320, 128, 3, t), (128, 128, 3, c), (128, 128, 3, c)),
            _seq((160, 64, 3, t), (64, 64, 3, c), (64, 64, 3, c)),
        ])
        self.output_block = nn.Sequential(_Block(80, 32, 3), nn.Conv2d(32, 3, 1), nn.Sigmoid())
        # "bf16x3" (default) meets the L-inf <= 1e-3 parity bound; "bf16" is the raw-speed mode
        self.precision = precision or os.environ.get("MF_PRECISION", "bf16x3")
        if self.precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {self.precision!r}")
        self._handle = None
        self._handle_device = None
        # lipreal.py:43-53 builds the model in the session's own process and then says `.to('cuda')`: on a multi-GPU node this is where the process takes its
        # GPU (placement.py: least-loaded, before the HIP runtime is up; a no-op with one GPU, under torch.distributed.run, or with MF_PLACEMENT=0)
        placement.charge_session(self)

    # ---- handle lifetime: any change to the parameters invalidates the packed weights ----------
    def _drop_handle(self):
        if getattr(self, "_handle", None):
            _lib.lib().mf_wav2lip_destroy(self._handle)
        self._handle = None
        self._handle_device = None

    def load_state_dict(self, *a, **kw):
        self._drop_handle()
        return super().load_state_dict(*a, **kw)

    def _apply(self, fn, *a, **kw):
        self._drop_handle()
        return super()._apply(fn, *a, **kw)

    def __del__(self):
        try:
            self._drop_handle()
        except Exception:
            pass

    def _ensure_handle(self, device):
        if self._handle is not None and self._handle_device == device:
            return self._handle
        self._drop_handle()
        _lib.init_device(device.index if device.index is not None else torch.cuda.current_device())
        items = [(k, v.detach().to("cpu", torch.float32).contiguous())
                 for k, v in self.state_dict().items() if not k.endswith("num_batches_tracked")]
        arr = (_lib.MfTensor * len(items))()
        keep = []
        for i, (k, v) in enumerate(items):
            name = k.encode()
            keep.append((name, v))
            arr[i].name = name
            arr[i].data = v.data_ptr()
            arr[i].ndim = v.dim()
            for d in range(v.dim()):
                arr[i].shape[d] = v.shape[d]
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().mf_wav2lip_create(arr, len(items), _lib.PRECISIONS[self.precision], C.byref(h)),
                       "wav2lip_create")
        self._handle, self._handle_device = h.value, device
        return self._handle

    # ---- wav2lip.py:87-125 -----------------------------------------------------------------------
    def forward(self, audio_sequences, face_sequences):
        if self.training:
            raise RuntimeError("the MI355X Wav2Lip generator is inference-only: call .eval() (lipreal.py:53)")
        if not face_sequences.is_cuda or not audio_sequences.is_cuda:
            raise RuntimeError("Wav2Lip.forward needs HIP device tensors (model.to('cuda')); no CPU path exists here")
        B = audio_sequences.size(0)
        five_d = face_sequences.dim() > 4
        if five_d:  # (B, T, 1, 80, 16) and (B, 6, T, 96, 96): fold T into the batch, wav2lip.py:92-94
            audio_sequences = torch.cat([audio_sequences[:, i] for i in range(audio_sequences.size(1))], dim=0)
            face_sequences = torch.cat([face_sequences[:, :, i] for i in range(face_sequences.size(2))], dim=0)
        h = self._ensure_handle(face_sequences.device)
        x = ops.wav2lip_forward(h, audio_sequences, face_sequences)
        if five_d:
            x = torch.stack(torch.split(x, B, dim=0), dim=2)  # (B, C, T, H, W)
        return x

    def forward_u8(self, mel_batch, faces_u8):
        """Fused lipreal.py:115-126: uint8 BGR crops [B,96,96,3] -> fp32 frames [B,96,96,3] (= pred*255)."""
        if self.training:
            raise RuntimeError("the MI355X Wav2Lip generator is inference-only: call .eval()")
        h = self._ensure_handle(faces_u8.device)
        return ops.wav2lip_forward_u8(h, mel_batch, faces_u8)

    def tune(self, batch):
        """Explicit launch-configuration warm-up (mf_wav2lip_tune): times every implicit-GEMM layer at this batch size on the buffers of the last
        forward at that size and keeps the fastest.  A forward itself never measures: it uses the tuning table (MF_TUNE_CACHE or the one shipped
        beside the library) or the cost model."""
        if self._handle is None:
            raise RuntimeError("Wav2Lip.tune: run one forward first")
        dev = self._handle_device
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mf_wav2lip_tune(self._handle, int(batch), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "wav2lip_tune")

    def read_tap(self, name, batch):
        """Intermediate activation of the last forward as fp32 NCHW (parity tests)."""
        shapes = {"audio_embedding": (512, 1, 1)}
        enc = [(16, 96), (32, 48), (64, 24), (128, 12), (256, 6), (512, 3), (512, 1)]
        dec = [(512, 1), (512, 3), (512, 6), (384, 12), (256, 24), (128, 48), (64, 96)]
        for i, (c, s) in enumerate(enc):
            shapes[f"face_encoder_blocks.{i}"] = (c, s, s)
        for i, (c, s) in enumerate(dec):
            shapes[f"face_decoder_blocks.{i}"] = (c, s, s)
        dev = self._handle_device
        out = torch.empty((batch,) + shapes[name], dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().mf_wav2lip_read_tap(self._handle, name.encode(), out.data_ptr(), batch,
                                                      C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                       "wav2lip_read_tap")
        return out
