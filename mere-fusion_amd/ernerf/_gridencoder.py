"""`import _gridencoder as _backend` (ernerf/gridencoder/grid.py:10): forward of the multi-resolution grid encoder."""
import ctypes as C

import torch

from . import backend as B

_offsets_cache = {}


def _host_offsets(offsets, L):
    """The level offsets never change after GridEncoder.__init__ (grid.py:108-123): one D2H copy per tensor."""
    key = (offsets.data_ptr(), L)
    hit = _offsets_cache.get(key)
    if hit is None:
        if offsets.dtype != torch.int32:
            raise RuntimeError("offsets must be int32 (grid.py:121)")
        host = offsets.detach().cpu().contiguous()
        hit = (host, (C.c_int * (L + 1))(*host.tolist()))
        _offsets_cache[key] = hit
    return hit[1]


def grid_encode_forward(inputs, embeddings, offsets, outputs, B_, D, C_, L, S, H, dy_dx, gridtype, align_corners):
    """grid.py:49 -> gridencoder.cu:404-440.  outputs: [L, B, C] (grid.py:42)."""
    if dy_dx is not None:
        raise RuntimeError("_gridencoder.grid_encode_forward: dy_dx (input gradients) is a training feature; pass None")
    if embeddings.dtype != torch.float32:
        raise RuntimeError("embeddings must be float32 (the half path of grid.py:36-39 only triggers under autocast)")
    B.call("mf_grid_encode_forward", B.f32(inputs, "inputs"), B.f32(embeddings, "embeddings"), _host_offsets(offsets, int(L)),
           B.f32(outputs, "outputs"), int(B_), int(D), int(C_), int(L), float(S), int(H), int(gridtype), int(bool(align_corners)), 0, B.stream())


def grid_encode_forward_blc(inputs, embeddings, offsets, outputs, B_, D, C_, L, S, H, gridtype, align_corners):
    """MI355X-native variant: writes [B, L*C] directly (what grid.py:52 permutes the reference output into)."""
    B.call("mf_grid_encode_forward", B.f32(inputs, "inputs"), B.f32(embeddings, "embeddings"), _host_offsets(offsets, int(L)),
           B.f32(outputs, "outputs"), int(B_), int(D), int(C_), int(L), float(S), int(H), int(gridtype), int(bool(align_corners)), 1, B.stream())


def grid_encode_backward(*a, **k):
    raise RuntimeError("_gridencoder.grid_encode_backward: training is outside the MI355X inference path")
