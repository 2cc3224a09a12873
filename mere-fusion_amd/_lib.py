"""ctypes binding of include/merefusion.h (the C ABI of libmerefusion_hip.so).

There is no CPU fallback anywhere in this package: if the library is missing or a call fails,
a RuntimeError carrying mf_last_error() is raised.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmerefusion_hip.so")

MF_PREC_BF16 = 0
MF_PREC_BF16X3 = 1
PRECISIONS = {"bf16": MF_PREC_BF16, "bf16x3": MF_PREC_BF16X3}


class MfTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int), ("shape", C.c_int64 * 4)]


class MfConv2dDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "cin", "cout", "kh", "kw", "stride_h", "stride_w", "pad_h", "pad_w", "transposed",
        "output_padding", "residual", "act", "in_h", "in_w")]


# every symbol include/merefusion.h declares: (restype, argtypes)
SIGNATURES = {
    "mf_init": (C.c_int, [C.c_int]),
    "mf_last_error": (C.c_char_p, []),
    "mf_abi_version": (C.c_int, []),
    "mf_wav2lip_create": (C.c_int, [C.POINTER(MfTensor), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mf_wav2lip_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mf_wav2lip_forward_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mf_wav2lip_read_tap": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mf_wav2lip_num_launches": (C.c_int, [C.c_void_p, C.c_int]),
    "mf_wav2lip_launch_info": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                        C.POINTER(C.c_double)]),
    "mf_wav2lip_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.POINTER(C.c_float), C.c_void_p]),
    "mf_wav2lip_destroy": (None, [C.c_void_p]),
    "mf_conv2d_create": (C.c_int, [C.POINTER(MfConv2dDesc)] + [C.c_void_p] * 6 + [C.c_int, C.POINTER(C.c_void_p)]),
    "mf_conv2d_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mf_conv2d_out_shape": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mf_conv2d_time": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]),
    "mf_conv2d_destroy": (None, [C.c_void_p]),
    "mf_whisper_create": (C.c_int, [C.POINTER(MfTensor), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mf_whisper_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mf_whisper_log_mel": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mf_whisper_encode_audio": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mf_whisper_destroy": (None, [C.c_void_p]),
    "mf_melspec": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "mf_melspec_frames": (C.c_int, [C.c_int]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is not built. Run `python -m mere_fusion_amd.build` (needs hipcc); "
                "this package has no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().mf_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"merefusion {what} failed (status {rc}): {msg}")


_inited = set()


def init_device(index):
    if index not in _inited:
        check(lib().mf_init(int(index)), "mf_init")
        _inited.add(index)
