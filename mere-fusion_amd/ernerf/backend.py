"""Shared plumbing of the ER-NeRF extension shims: tensor checks and the raw-pointer calls into the C ABI.

The reference's pybind functions take torch tensors and write their outputs in place (raymarching.cu:147-155 etc.); the
shims keep exactly that contract.  There is no CPU path: a CPU tensor raises, as the CUDA extension would (CHECK_CUDA)."""
import ctypes as C

import torch

from .. import _lib


def _ptr(t, dtype, name):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")            # CHECK_CUDA, raymarching.cu:13
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")      # CHECK_CONTIGUOUS, raymarching.cu:14
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype} (got {t.dtype}); the wrappers cast with custom_fwd(cast_inputs=float32)")
    return C.c_void_p(t.data_ptr())


def f32(t, name):
    return _ptr(t, torch.float32, name)


def i32(t, name):
    return _ptr(t, torch.int32, name)


def u8(t, name):
    return _ptr(t, torch.uint8, name)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(fn_name, *args):
    lib = _lib.lib()
    _lib.check(getattr(lib, fn_name)(*args), fn_name)
