"""`import _freqencoder as _backend` (ernerf/freqencoder/freq.py:10)."""
from . import backend as B


def freq_encode_forward(inputs, B_, input_dim, degree, output_dim, outputs):
    """freq.py:29 -> freqencoder.cu:96-105."""
    B.call("mf_freq_encode_forward", B.f32(inputs, "inputs"), int(B_), int(input_dim), int(degree), int(output_dim), B.f32(outputs, "outputs"),
           B.stream())


def freq_encode_backward(*a, **k):
    raise RuntimeError("_freqencoder.freq_encode_backward: training is outside the MI355X inference path")
