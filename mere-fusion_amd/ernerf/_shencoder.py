"""`import _shencoder as _backend` (ernerf/shencoder/sphere_harmonics.py:10)."""
from . import backend as B


def sh_encode_forward(inputs, outputs, B_, input_dim, degree, dy_dx):
    """sphere_harmonics.py:32 -> shencoder.cu:405-419."""
    if int(input_dim) != 3:
        raise RuntimeError("SH encoder: input_dim must be 3 (sphere_harmonics.py:63)")
    if dy_dx is not None:
        raise RuntimeError("_shencoder.sh_encode_forward: dy_dx is a training feature; pass None")
    B.call("mf_sh_encode_forward", B.f32(inputs, "inputs"), B.f32(outputs, "outputs"), int(B_), int(degree), B.stream())


def sh_encode_backward(*a, **k):
    raise RuntimeError("_shencoder.sh_encode_backward: training is outside the MI355X inference path")
