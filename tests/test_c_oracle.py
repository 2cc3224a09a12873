"""CPU: the plain-C float64 direct convolution (oracle/conv_ref.c) agrees with the torch restatement
and with the reference's recorded outputs -- an arithmetic check independent of torch's conv kernels."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import geometry_cases as G
from conftest import ROOT


@pytest.fixture(scope="module")
def cref():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)
    return C.CDLL(os.path.join(ROOT, "oracle", "libconvref.so"))


SMALL = [c for c in G.CASES if c["cin"] * c["cout"] * c["h"] * c["w"] * c["k"] ** 2 <= 400e6]


@pytest.mark.parametrize("case", SMALL, ids=[c["name"] for c in SMALL])
def test_c_oracle_vs_reference_golden(cref, conv_golden, case):
    p = {k: np.ascontiguousarray(v.numpy()) for k, v in G.case_params(case).items()}
    x = np.ascontiguousarray(G.case_input(case).numpy())
    ref = conv_golden[f"y/{case['name']}"]
    y = np.zeros(ref.shape, np.float64)
    f = lambda a: a.ctypes.data_as(C.c_void_p)
    N, ci, H, W_ = x.shape
    if case["transposed"]:
        cref.ref_conv_transpose2d(f(x), f(p["weight"]), f(p["bias"]), f(y), N, ci, H, W_, case["cout"], case["k"],
                                  case["stride"], case["pad"], case["outpad"])
    else:
        sh, sw = (case["stride"], case["stride"]) if isinstance(case["stride"], int) else case["stride"]
        cref.ref_conv2d(f(x), f(p["weight"]), f(p["bias"]), f(y), N, ci, H, W_, case["cout"], case["k"], case["k"],
                        sh, sw, case["pad"], case["pad"])
    cref.ref_bn_res_relu.argtypes = [C.c_void_p] * 5 + [C.c_double, C.c_void_p, C.c_int, C.c_int, C.c_int]
    cref.ref_bn_res_relu(f(y), f(p["gamma"]), f(p["beta"]), f(p["mean"]), f(p["var"]), 1e-5,
                         f(x) if case["residual"] else None, N, case["cout"], ref.shape[2] * ref.shape[3])
    np.testing.assert_allclose(y, ref, rtol=2e-5, atol=2e-5)
