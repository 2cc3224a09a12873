"""`import _gridencoder as _backend` (ernerf/gridencoder/grid.py:10): forward of the multi-resolution grid encoder."""
import ctypes as C

import torch

from . import backend as B

def _host_offsets(offsets, L):
    """The level offsets never change after GridEncoder.__init__ (grid.py:108-123): one D2H copy per tensor OBJECT, kept on the tensor
    itself (a cache keyed by data_ptr would alias a freed tensor's address once the allocator reuses it)."""
    hit = getattr(offsets, "_mf_host_offsets", None)
    if hit is None or hit[0] != int(L) or hit[1] != offsets._version:
        if offsets.dtype != torch.int32:
            raise RuntimeError("offsets must be int32 (grid.py:121)")
        host = offsets.detach().cpu().contiguous()
        hit = (int(L), offsets._version, (C.c_int * (L + 1))(*host.tolist()))
        try:
            offsets._mf_host_offsets = hit
        except AttributeError:      # (a tensor subclass without a __dict__: just do the copy every call)
            pass
    return hit[2]


def grid_encode_forward(inputs, embeddings, offsets, outputs, B_, D, C_, L, S, H, dy_dx, gridtype, align_corners):
    """grid.py:49 -> gridencoder.cu:404-440.  outputs: [L, B, C] (grid.py:42).

    Under autocast the reference casts the embeddings of an even-C encoder to half and allocates half outputs (grid.py:36-42; app.py:363
    forces fp16 for ernerf, and the torso's tiled grid has level_dim 2, network.py:162).  The kernel here interpolates in fp32: half
    embeddings are widened exactly, the fp32 result is rounded ONCE into the caller's half `outputs` -- the reference's half accumulator
    rounds after every corner, so the two agree to one fp16 ulp of the running sum, with this side the more accurate."""
    if dy_dx is not None:
        raise RuntimeError("_gridencoder.grid_encode_forward: dy_dx (input gradients) is a training feature; pass None")
    if embeddings.dtype not in (torch.float32, torch.float16) or outputs.dtype != embeddings.dtype:
        raise RuntimeError(f"embeddings / outputs must both be float32 or float16 (grid.py:36-42), got {embeddings.dtype} / {outputs.dtype}")
    emb32 = embeddings if embeddings.dtype == torch.float32 else embeddings.float()
    out32 = outputs if outputs.dtype == torch.float32 else torch.empty(outputs.shape, dtype=torch.float32, device=outputs.device)
    B.call("mf_grid_encode_forward", B.f32(inputs, "inputs"), B.f32(emb32, "embeddings"), _host_offsets(offsets, int(L)),
           B.f32(out32, "outputs"), int(B_), int(D), int(C_), int(L), float(S), int(H), int(gridtype), int(bool(align_corners)), 0, B.stream())
    if out32 is not outputs:
        outputs.copy_(out32)


def grid_encode_forward_blc(inputs, embeddings, offsets, outputs, B_, D, C_, L, S, H, gridtype, align_corners):
    """MI355X-native variant: writes [B, L*C] directly (what grid.py:52 permutes the reference output into)."""
    B.call("mf_grid_encode_forward", B.f32(inputs, "inputs"), B.f32(embeddings, "embeddings"), _host_offsets(offsets, int(L)),
           B.f32(outputs, "outputs"), int(B_), int(D), int(C_), int(L), float(S), int(H), int(gridtype), int(bool(align_corners)), 1, B.stream())


def grid_encode_backward(*a, **k):
    raise RuntimeError("_gridencoder.grid_encode_backward: training is outside the MI355X inference path")
